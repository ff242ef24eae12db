/* include/crazyara_hip.h must be a valid C99 header: compiled by tests/test_capi_exports.py with -std=c99 -Wall -Werror -pedantic.
 * Also pins the layout of the structs a C consumer passes by pointer. */
#include "crazyara_hip.h"

#include <stddef.h>

typedef char assert_settings_has_no_holes_at_the_end[(sizeof(mi_search_settings) % 4 == 0) ? 1 : -1];
typedef char assert_stats_double_aligned[(offsetof(mi_search_stats, seconds) % 8 == 0) ? 1 : -1];

size_t cra_sizeof_search_settings(void) { return sizeof(mi_search_settings); }
size_t cra_sizeof_search_stats(void) { return sizeof(mi_search_stats); }
size_t cra_offsetof_settings_virtual_offset_strength(void) { return offsetof(mi_search_settings, virtual_offset_strength); }
size_t cra_offsetof_settings_version_minor(void) { return offsetof(mi_search_settings, version_minor); }
size_t cra_offsetof_stats_depth_max(void) { return offsetof(mi_search_stats, depth_max); }
size_t cra_sizeof_selfplay_settings(void) { return sizeof(mi_selfplay_settings); }
size_t cra_sizeof_selfplay_stats(void) { return sizeof(mi_selfplay_stats); }
size_t cra_offsetof_selfplay_seed(void) { return offsetof(mi_selfplay_settings, seed); }
size_t cra_offsetof_selfplay_stats_wins(void) { return offsetof(mi_selfplay_stats, wins); }

/* typed function pointers: the declarations are complete prototypes with the expected parameter lists */
const char* (*const cra_p_last_error)(void) = mi_last_error;
mi_net* (*const cra_p_net_create)(const char*, int, int, const char*) = mi_net_create;
int (*const cra_p_net_predict)(mi_net*, const float*, float*, float*, float*) = mi_net_predict;
void (*const cra_p_net_destroy)(mi_net*) = mi_net_destroy;
void* (*const cra_p_host_alloc)(size_t) = mi_host_alloc;
int (*const cra_p_search_run)(mi_search*, unsigned, unsigned, int, mi_search_stats*) = mi_search_run;
mi_selfplay* (*const cra_p_selfplay_create)(mi_search*, mi_search*, const mi_selfplay_settings*, int, const char*, int, mi_traindata*) = mi_selfplay_create;
int (*const cra_p_selfplay_play)(mi_selfplay*, int, int) = mi_selfplay_play;
size_t cra_sizeof_go_limits(void) { return sizeof(mi_go_limits); }
size_t cra_offsetof_go_limits_move_overhead(void) { return offsetof(mi_go_limits, move_overhead); }
