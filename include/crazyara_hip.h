/*
 * crazyara_hip.h -- C ABI of the MI355X-native evaluator (libcrazyara_hip.so).
 *
 * This is the drop-in boundary for CrazyAra's NN plugin surface: a ~100-line `HipAPI : NeuralNetAPI` shim
 *  binds exactly these entry points.  Every function cites the reference interface it replaces
 * (paths relative to the reference repository root).  Plain pointers and sizes only; no C++/torch types.
 *
 * Error model: functions returning int give 0 on success, non-zero on failure; functions returning a pointer give
 * NULL on failure; mi_last_error() then holds a thread-local message.  (The reference throws from constructors --
 * engine/src/nn/neuralnetapi.cpp:65-70 -- and has no error channel in predict(); the shim re-throws / aborts.)
 */
#ifndef CRAZYARA_HIP_H
#define CRAZYARA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------------------------------------- */
const char* mi_last_error(void);
const char* mi_version(void);
/* number of visible HIP devices (UCI options First_Device_ID / Last_Device_ID, engine/src/uci/optionsuci.cpp:114-127) */
int mi_device_count(void);

/* Pinned host buffers: replaces cudaMallocHost / cudaFreeHost in NeuralNetAPIUser
 * (engine/src/nn/neuralnetapiuser.cpp:50-60,80-88). */
void* mi_host_alloc(size_t bytes);
void mi_host_free(void* p);

/* ------------------------------------------------------------------------------------------------------------------
 * Network handle == one NeuralNetAPI instance (engine/src/nn/neuralnetapi.h:148-311)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_net mi_net;

/* TensorrtAPI::TensorrtAPI(deviceID, batchSize, modelDirectory, strPrecision) + initialize()
 * (engine/src/nn/tensorrtapi.cpp:43-63; neuralnetapi.cpp:93-99).  model_dir: directory holding a "*.cranet" file
 * or, as the reference's model directories do, an "*.onnx" file (chosen like get_onnx_model_name(), neuralnetapi.cpp:57-73:
 * "-bsize-<B>" first, else the one without "-bsize-"; a .cranet wins over an .onnx), or a direct path to either.
 * ONNX files are read in place (the graphs of the reference's model zoo, see csrc/nn/onnx_import.h), the "-v<maj>.<min>"
 * part of the name is the input-representation version (read_version_from_string, neuralnetapi.cpp:194-227).
 * precision: "float32" | "float16" (UCI option Precision, optionsuci.cpp:143-147) | "float16x3" (float activations, every dense contraction
 * as three f16 MFMAs on hi/lo split operands: logits within ~1e-5 of fp32 at several times the float32 rate; every net family) | "fp8" (also "float8"; "int8" -- the third value of the
 * reference's option, TensorRT INT8 with a calibration cache, tensorrtapi.cpp:229-248 -- is accepted as a name for it): float16 with OCP
 * e4m3 operands in the two GEMMs of every residual block (f32 accumulation, per-row power-of-two weight scales, no calibration file; the
 * residual stream, stem and heads stay f16).  256-channel bottleneck (RISE) nets only; error against fp32 about 2^7 times float16's
 * (DESIGN 4.3).  Other models: RuntimeError, as an unsupported precision is in the reference. */
mi_net* mi_net_create(const char* model_dir, int device_id, int batch_size, const char* precision);
/* Precision "int8" -- the reference's calibrated INT8 mode (TensorRT INT8 with an Int8EntropyCalibrator2 over the engine's
 * ChessBatchStream positions, engine/src/nn/tensorrtapi.cpp:334-360, environments/chess_related/chessbatchstream.cpp:44-94) -- needs one
 * calibration pass per model, kept beside it as <model file>.int8calib (TensorRT keeps its calibration cache the same way):
 * mi_net_calibrate_int8 runs n_boards positions (float planes [n][C][8][8], what ChessBatchStream::getBatch hands the calibrator) through
 * the float16 layer kernels on `device_id` and records, per bottleneck block, the largest |value| of the stream in front of it and of its
 * depthwise output; mi_net_create(..., "int8") then quantises activations with those steps (one per tensor), weights per output row, and
 * runs both GEMMs of every block on v_mfma_i32_32x32x32_i8.  planes == NULL: the library's default calibration positions -- the plies of
 * the reference's own calibration games (chessbatchstream.cpp:44-94: 232 crazyhouse / 104 chess plies, kept as data in
 * crazyara_amd/data/opening_games.json beside the library, CRA_DATA_DIR names another directory), encoded with the plane layout the
 * model's input shape and file-name version select.  mi_net_create fails with a message that names this call when the file is
 * missing.  mi_net_has_int8_calibration: 1 / 0, -1 on error. */
int mi_net_calibrate_int8(const char* model_dir, int device_id, const float* planes, int n_boards);
int mi_net_has_int8_calibration(const char* model_dir);
void mi_net_destroy(mi_net* net);

/* The weight quantiser of precision "fp8" (host only): float -> OCP e4m3fn byte, round to nearest even, |v| >= 448 clamps to +-448,
 * NaN -> 0x7f.  Exposed so that the CPU test suite can pin it against an independent e4m3 conversion. */
int mi_e4m3_from_float(float v);
/* The weight quantiser of the cross-term images of precision "float16p8" (host only): float -> e5m2 byte (f16's exponent field, two
 * mantissa bits), round to nearest even, subnormals kept, beyond the range the largest finite value.  Exposed for the same reason. */
int mi_e5m2_from_float(float v);

/* Offline form of the same import: parse the ONNX file and write it as a .cranet container (what TensorrtAPI caches as a
 * serialized engine next to the ONNX, tensorrtapi.cpp:297-332).  Host only, no GPU needed.  0 on success. */
int mi_onnx_to_cranet(const char* onnx_path, const char* cranet_path);

/* nnDesign + derived getters (neuralnetapi.cpp:81-91, neuralnetapi.h:252-299):
 * in_shape = {batch, C, 8, 8}; nb_policy = get_nb_policy_values(); nb_aux = get_nb_auxiliary_outputs();
 * version = get_version() as make_version(maj,min,0) (engine/src/version.h:53-55); game_phase = get_game_phase(). */
int mi_net_design(const mi_net* net, int in_shape[4], int* nb_policy, int* nb_aux, int* version, int* game_phase);
const char* mi_net_model_name(const mi_net* net);        /* get_model_name(), neuralnetapi.cpp:116-119 */
double mi_net_flops_per_position(const mi_net* net);     /* 2*MACs of the loaded layer list (for roofline reporting) */

/* NeuralNetAPI::predict(float* inputPlanes, float* valueOutput, float* probOutputs, float* auxiliaryOutputs)
 * (neuralnetapi.h:230-237; behaviour of tensorrtapi.cpp:195-237): whole fixed batch, fp32 NCHW host planes,
 * blocking; value after tanh, probs = softmax over all nb_policy entries, aux only if nb_aux > 0. */
int mi_net_predict(mi_net* net, const float* in_planes, float* value, float* probs, float* aux);

/* Asynchronous split of predict(): the MI355X restatement of "two SearchThreads per GPU hide the blocking call"
 * (engine/src/searchthread.cpp:403-416; optionsuci.cpp:182): submit enqueues H2D + forward + D2H on the net's
 * side stream, wait blocks until the host buffers are valid. */
int mi_net_submit(mi_net* net, const float* in_planes, float* value, float* probs, float* aux);
int mi_net_wait(mi_net* net);
/* predict / submit with EVERY buffer of the call in pinned host memory (mi_host_alloc, or any hipHostMalloc / hipHostRegister
 * memory -- what NeuralNetAPIUser owns under its TENSORRT branch, neuralnetapiuser.cpp:50-60) issue no copy commands: the forward's
 * first kernel reads the planes and its last kernels write value / probabilities / aux in place over PCIe (one queue, no hand-over
 * to the DMA engines).  Pageable buffers are copied as before.  Returns 1 if the last predict / submit of this net took that path. */
int mi_net_last_submit_zero_copy(const mi_net* net);

/* Device-resident path (no PCIe): pointers to the buffers the captured forward reads/writes.
 * d_planes [B][C][64] float, d_value [B], d_probs [B][nb_policy], d_logits [B][nb_policy] (pre-softmax policy_out),
 * d_aux [B][4] or NULL.  Any out-pointer may be NULL. */
int mi_net_device_buffers(mi_net* net, float** d_planes, float** d_value, float** d_probs, float** d_logits, float** d_aux);
/* d_logits is test / analysis output (predict's contract is the softmaxed vector): forwards write it only after mi_net_keep_logits(net, 1)
 * (the one-launch head keeps the logits in LDS otherwise; nets whose heads run as separate launches always have it). */
int mi_net_keep_logits(mi_net* net, int on);
/* Test hook of the one-launch bottleneck tower (Precision float16 / fp8 on 256-channel RISE nets): from this call on every forward also
 * stores the f16 residual stream in front of the first block and behind every block.  Returns the device buffer
 * [*n_tiles = blocks + 1][batch][64][256] f16 (NULL + mi_last_error when the net does not run that kernel).  What a parity test needs to
 * check a reduced-precision mode block by block (tests/test_fp8.py) -- the reference has no counterpart. */
void* mi_net_block_dump(mi_net* net, int* n_tiles);
int mi_net_forward_device(mi_net* net);          /* hipGraph replay on the net stream, asynchronous */
int mi_net_sync(mi_net* net);                    /* hipStreamSynchronize(net stream) */
void* mi_net_stream(mi_net* net);                /* the hipStream_t */

/* Run `iters` forwards (graph replays) bracketed by hipEvents on the net's own stream; *ms_total = elapsed.
 * This is the NN-only shape of CrazyAra::inference (engine/src/uci/crazyara.cpp:156-181) minus the PCIe copies. */
int mi_net_time_forward(mi_net* net, int iters, float* ms_total);
/* Per-launch timing: each op of the forward launched un-graphed between two hipEvents, `iters` times.
 * names: op_count pointers to static strings; ms: op_count accumulated milliseconds (sum over iters). */
int mi_net_op_count(const mi_net* net);
int mi_net_time_ops(mi_net* net, int iters, const char** names, float* ms);

/* ------------------------------------------------------------------------------------------------------------------
 * Environment: position, legal moves, terminal rules (State interface of engine/src/state.h:287-509 as implemented by
 * BoardState, engine/src/environments/chess_related/boardstate.cpp:42-277, on top of the fork's Position)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_pos mi_pos;
/* modes = the reference's build flavours (engine/CMakeLists.txt:28-64) */
enum { MI_MODE_CRAZYHOUSE = 0, MI_MODE_CHESS = 1, MI_MODE_LICHESS = 2 };
/* TerminalType, engine/src/state.h */
enum { MI_TERMINAL_LOSS = 0, MI_TERMINAL_DRAW = 1, MI_TERMINAL_WIN = 2, MI_TERMINAL_CUSTOM = 3, MI_TERMINAL_NONE = 4 };

/* BoardState::set(fen, isChess960, variant) (boardstate.cpp:71-75); variant = UCI_Variant name ("crazyhouse", "chess",
 * "3check", "kingofthehill", ...; boardstate.h:296-320).  fen == NULL or "" -> start position (BoardState::init). */
mi_pos* mi_pos_create(const char* fen, int is_chess960, const char* variant);
mi_pos* mi_pos_clone(const mi_pos* pos);                               /* BoardState::clone, boardstate.cpp:255-258 */
void mi_pos_destroy(mi_pos* pos);
int mi_pos_fen(const mi_pos* pos, char* buf, int cap);                 /* BoardState::fen; returns length or -1 */
int mi_pos_side_to_move(const mi_pos* pos);                            /* 0 white, 1 black */
int mi_pos_legal_moves(const mi_pos* pos, uint32_t* moves, int cap);   /* BoardState::legal_actions; returns count */
uint32_t mi_pos_uci_to_move(const mi_pos* pos, const char* uci);       /* uci_to_action; 0 if not legal */
int mi_pos_move_to_uci(const mi_pos* pos, uint32_t move, char* buf, int cap);  /* StateConstants::action_to_uci */
/* SAN in the reference's dialect (pgn_move, board.cpp:277-359): "Nbd7", "exd5", "e8Q", "P@e4", "O-O", '+' on check */
int mi_pos_move_to_san(const mi_pos* pos, uint32_t move, char* buf, int cap);
int mi_pos_do_move(mi_pos* pos, uint32_t move);                        /* do_action */
int mi_pos_terminal(const mi_pos* pos);                                /* is_terminal(legal count) -> MI_TERMINAL_* */
/* Board::get_phase (environments/chess_related/board.cpp:540-587): definition 0 = lichess (0 opening, 1 middlegame, 2 endgame; majors and
 * minors, sparse back rank, mixedness of the Divider, board.cpp:446-538), 1 = movecount (num_phases slices of a 42.85-move game); -1 on error */
int mi_pos_game_phase(const mi_pos* pos, int num_phases, int definition);
int mi_pos_number_repetitions(const mi_pos* pos);
int mi_pos_insufficient_material(const mi_pos* pos);                  /* Board::draw_by_insufficient_material (board.cpp:175-221) */
int mi_pos_plies_from_null(const mi_pos* pos);                        /* State::steps_from_null (boardstate.h) */
int mi_pos_in_check(const mi_pos* pos);                                /* 1 if the side to move is in check (Position::checkers) */
unsigned long long mi_pos_perft(const mi_pos* pos, int depth);
const char* mi_chess960_start_fen(int scharnagl_index);                /* deterministic stand-in for chess960fen() */

/* ------------------------------------------------------------------------------------------------------------------
 * Input planes: board_to_planes (engine/src/environments/chess_related/inputrepresentation.cpp:628-680)
 * ---------------------------------------------------------------------------------------------------------------- */
int mi_planes_layout(int mode, int version_major);        /* layout id used below */
/* the same with the minor number: MI_MODE_CHESS 2.7 / 2.8 (board_to_planes_chess_v_2_7 / _2_8, inputrepresentation.cpp:503-533) */
int mi_planes_layout_minor(int mode, int version_major, int version_minor);
int mi_planes_channels(int layout);                       /* NB_CHANNELS_TOTAL of that layout */
/* host builder: out[C*64] floats NCHW.  repetitions < 0 -> Board::number_repetitions() */
int mi_pos_planes(const mi_pos* pos, int layout, int normalize, int repetitions, float* out);
/* compact 192-byte descriptor of the position (struct BoardDesc, crazyara_amd/csrc/chess/planes.h) */
int mi_pos_desc(const mi_pos* pos, void* desc192);
/* descriptor for one layout: also fills the legal-move features (check-giving moves, mobility) when the layout reads them
 * (chess 2.7 / 2.8); for every other layout identical to mi_pos_desc */
int mi_pos_desc_for(const mi_pos* pos, int layout, void* desc192);
/* host builder from descriptors: out[n][C*64] floats (what a CPU NeuralNetAPI behind the same leaf collector is fed) */
int mi_planes_from_descs_host(const void* descs, int n, int layout, int normalize, float* out);
/* GPU builder: n host descriptors -> d_planes (device) [n][C][64] float, blocking */
int mi_planes_from_descs_device(const void* descs_host, int n, int layout, int normalize, float* d_planes, int device_id);
/* predict() fed by descriptors instead of float planes: H2D of 192 B/board, planes built on the GPU straight into the
 * net's input tensor, forward, D2H.  Only the first n_valid slots are rebuilt (trailing slots keep stale data, as in
 * engine/src/searchthread.cpp:407-411).  submit/wait semantics as mi_net_submit. */
int mi_net_submit_boards(mi_net* net, const void* descs_host, int n_valid, int layout, float* value, float* probs, float* aux);
/* The same for a search that knows which entries it will read: slot s passes the policy indices of its position's legal moves,
 * idx[s * stride .. + cnt[s]), and gets back gathered[s * stride + j] = probs[s][idx[s * stride + j]] -- what
 * Node::set_probabilities_for_moves (node.cpp:961-979) picks out of the probability vector -- instead of all nb_policy floats
 * (a batch of 256 crazyhouse boards: ~170 KB instead of 5.3 MB).  No copy commands are issued: descs_host, idx, cnt, value,
 * gathered and aux must be mi_host_alloc memory, the kernels read and write them in place.  mi_net_wait as for mi_net_submit. */
int mi_net_submit_boards_gathered(mi_net* net, const void* descs_host, int n_valid, int layout, const unsigned short* idx,
                                  const unsigned* cnt, unsigned stride, float* value, float* gathered, float* aux);

/* ------------------------------------------------------------------------------------------------------------------
 * Policy map (engine/src/environments/chess_related/outputrepresentation.cpp, policymaprepresentation.h)
 * ---------------------------------------------------------------------------------------------------------------- */
int mi_policy_nb_labels(int mode);                         /* StateConstants::NB_LABELS, boardstate.h:51-60 */
int mi_policy_nb_policy_map(int mode);                     /* NB_LABELS_POLICY_MAP, boardstate.h:61-63 */
const char* mi_policy_label(int mode, int idx, int mirrored);   /* LABELS / LABELS_MIRRORED */
int mi_policy_flat_plane_idx(int mode, int idx);           /* FLAT_PLANE_IDX[idx] */
/* MV_LOOKUP / MV_LOOKUP_MIRRORED as used by Node::set_probabilities_for_moves (engine/src/node.cpp:961-979):
 * index of `move` in the policy vector, mirrored table iff black to move.  -1 if the move has no label. */
int mi_pos_policy_index(const mi_pos* pos, uint32_t move, int mode, int is_policy_map);

/* ------------------------------------------------------------------------------------------------------------------
 * MCTS leaf collection (engine/src/searchthread.cpp:164-449, engine/src/node.{h,cpp}; driven like
 * MCTSAgent::evaluate_board_state -> run_mcts_search, engine/src/agents/mctsagent.cpp:292-362)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_search mi_search;
typedef struct mi_search_settings {        /* SearchSettings (engine/src/agents/config/searchsettings.h:51-99) */
    int batch_size;                        /* informational; the evaluator lanes fix the real batch */
    float cpuct_init, cpuct_base;          /* Centi_CPuct_Init / CPuct_Base */
    float node_policy_temperature;         /* Centi_Node_Temperature */
    int virtual_style;                     /* 0 LOSS, 1 VISIT, 2 OFFSET, 3 MIX (MCTS_Virtual_Style) */
    unsigned virtual_mix_threshold;
    double virtual_offset_strength;
    float q_value_weight, q_veto_delta;
    int mode;                              /* MI_MODE_* */
    int version_major;                     /* input representation of the net */
    int is_policy_map;
    int clone_keeps_last_moves;            /* -1 follow the build mode (board.cpp:106-108), 0 / 1 force */
    /* epsilon exploration (searchthread.cpp:124-185): 1 simulation in `counter` starts with a random playout / an unexplored
     * checking move from a random depth; 0 = off (mi_search_default_settings), reference UCI defaults 20 / 100
     * (Centi_Epsilon_Greedy 5, Centi_Epsilon_Checks 1).  seed: per-pool seed of the trees' generators (tree i uses seed + i). */
    int epsilon_greedy_counter, epsilon_checks_counter;
    unsigned seed;
    /* MCTS_Solver (optionsuci.cpp:129, default on): terminal backups prove WIN / LOSS / DRAW up the tree
     * (Node::solve_for_terminal, node.cpp:365-453); a proven root ends that tree's search. */
    int mcts_solver;
    /* Dirichlet noise on the root priors at the start of every search when epsilon > 0.009, then full expansion of the root
     * (mctsagent.cpp:311-316, node.cpp:950-954, blazeutil.h:113-124).  Defaults 0 / 0.2 (Centi_Dirichlet_Epsilon is 25 in RL builds). */
    float dirichlet_epsilon, dirichlet_alpha;
    int version_minor;                     /* chess input representation 2.7 / 2.8 (make_version<2,7,0>, <2,8,0>); 0 otherwise */
} mi_search_settings;
typedef struct mi_search_stats {
    unsigned long long nodes, nn_evals, batches, simulations;
    double seconds, depth_avg;
    unsigned depth_max;
} mi_search_stats;
/* evaluator callback lane: fill value[n] and probs[n*nb_policy] for n 192-byte descriptors; return 0 on success */
typedef int (*mi_eval_fn)(void* user, const void* descs, int n, float* value, float* probs);

void mi_search_default_settings(mi_search_settings* s);                 /* UCI defaults, optionsuci.cpp:66-219 */
/* lanes: net_a (required unless fn given), net_b optional second net instance on the same GPU -> collection of one half of
 * the trees overlaps evaluation of the other half.  With fn != NULL the nets are ignored and fn evaluates (batch, nb_policy given). */
mi_search* mi_search_create(const mi_search_settings* s, mi_net* net_a, mi_net* net_b, mi_eval_fn fn, void* user, int fn_batch, int fn_nb_policy);
/* one more evaluator lane (its own net handle: weights, stream, graph), to be added before the first position: with k lanes k
 * batches are in flight, like k SearchThreads of the reference each blocked in its own predict() (searchthread.cpp:403-416) */
int mi_search_add_lane(mi_search* sp, mi_net* net);
void mi_search_destroy(mi_search* sp);
int mi_search_add_position(mi_search* sp, const char* fen, int is_chess960, const char* variant);   /* returns tree id or -1 */
/* go with Simulations / Nodes limits per tree (searchthread.cpp:326-331); threads = host collector threads */
/* a move was played on the board of tree `tree` (own move or the opponent's reply): keep the subtree below it as the new tree
 * if it was searched, else restart from the new position -- MCTSAgent::apply_move_to_tree + get_root_node_from_tree
 * (mctsagent.cpp:130-164,230-247).  *kept = 1 if the subtree was reused.  uci must be a legal move of the tree's root position. */
int mi_search_apply_move(mi_search* sp, int tree, const char* uci, int* kept);
int mi_search_tree_fen(mi_search* sp, int tree, char* fen, int cap);   /* FEN of the tree's root position */
int mi_search_run(mi_search* sp, unsigned simulations, unsigned nodes, int threads, mi_search_stats* stats);
/* the same with a wall-clock limit (SearchLimits::movetime; the timer half of ThreadManager::stop_search_based_on_limits,
 * engine/src/manager/threadmanager.cpp:69-97): the searches also end movetime_ms after the start of the call.  At least one of the
 * three limits must be non-zero; batches in flight are applied before the call returns. */
int mi_search_run_timed(mi_search* sp, unsigned simulations, unsigned nodes, unsigned movetime_ms, int threads, mi_search_stats* stats);
/* Stop protocol.  mi_search_stop from ANOTHER thread ends the search that is running or announced as if its limits had been reached
 * (SearchThread::stop, engine/src/searchthread.cpp:109-112; MCTSAgent::stop, agents/mctsagent.cpp:364-373); the run call returns
 * with consistent trees.  mi_search_announce_go is what the commanding thread (the UCI loop on `go`, uci/crazyara.cpp:183-231) calls
 * BEFORE it hands mi_search_run / mi_search_run_timed to its search thread: a stop that arrives between the announcement and the
 * moment the search thread enters the run call is kept and ends that search at once (after the roots are evaluated, so that a best
 * move exists) -- it is sticky for ITS search and never reaches a later one.  A run that was not announced opens its own generation
 * on entry; with nothing announced or running a stop does nothing (MCTSAgent::stop: `if (!isRunning) return`).  Searches may overlap
 * at this level -- `stop` followed directly by `go` (or ponderhit) announces the next search while the previous run call is still
 * returning: a stop names every search announced or running at that moment, a run adopts the oldest announcement nobody has run yet,
 * and the previous run's exit touches neither (tests/test_mcts.py::test_go_announced_before_the_previous_run_returned_keeps_its_stop). */
int mi_search_announce_go(mi_search* sp);
/* Every mi_search_announce_go must be followed by exactly one run call -- or by mi_search_cancel_go when the commanding thread drops
 * the search before its run was entered (the search thread could not be started, a `stop` made the go pointless): the oldest
 * announcement no run has taken is withdrawn, together with any stop it has collected, so that the next run does not inherit either.
 * Returns 1 if an announcement was withdrawn, 0 if none was waiting, -1 on a null handle. */
int mi_search_cancel_go(mi_search* sp);
int mi_search_stop(mi_search* sp);
/* root statistics of one tree, children in the node's (prior-sorted) order: returns number of expanded children */
int mi_search_root_children(mi_search* sp, int tree, int cap, uint32_t* moves, uint32_t* visits, float* q, float* priors);
int mi_search_tree_info(mi_search* sp, int tree, unsigned* root_visits, unsigned* node_count, unsigned* allocated_nodes, float* root_value);
/* solver state of the root (NodeData::nodeType / endInPly / checkmateIdx, nodedata.h:88-121): node_type 0 WIN, 1 DRAW, 2 LOSS,
 * 6 UNSOLVED (NodeType, nodedata.h:40-52); end_in_ply = plies to the proven terminal; checkmate_idx = mating child or -1 */
int mi_search_root_solved(mi_search* sp, int tree, int* node_type, int* end_in_ply, int* checkmate_idx);
int mi_search_best_move(mi_search* sp, int tree, char* uci, int cap);
/* the whole Node::get_mcts_policy vector of the root (EvalInfo::policyProbSmall, one entry per expanded child in the order of
 * mi_search_root_children) and the Q value of the best move (EvalInfo::bestMoveQ); returns the number of entries or -1 */
int mi_search_root_policy(mi_search* sp, int tree, int cap, double* policy, float* best_move_q);
/* The movetime of a `go` with clock arguments: TimeManager::get_time_for_move (engine/src/manager/timemanager.cpp:50-103) with the
 * constants of constants.h:94-98 (expected game length 38, proportional system from move 35 on with 14 moves to go, increment
 * factor 0.7, safety buffer 30 x Move_Overhead) and randomMoveFactor 0.  side: 0 White, 1 Black; move_number = plies of the root / 2
 * (mctsagent.cpp:349).  Returns milliseconds; 0 = no time limit (infinite, or node / simulation / depth limits without movetime).
 * Feed the result to mi_search_run_timed. */
typedef struct mi_go_limits {              /* SearchLimits (engine/src/agents/config/searchlimits.h:32-61), the fields the function reads */
    long long movetime;
    unsigned long long nodes, simulations;
    int movestogo, depth;
    int time[2], inc[2];                   /* wtime / btime, winc / binc in ms */
    int move_overhead;                     /* Move_Overhead (optionsuci.cpp:135, default 20) */
    int infinite;
} mi_go_limits;
int mi_time_for_move(const mi_go_limits* limits, int side, int move_number);
/* what a UCI front end prints behind "info ... score" and "pv": EvalInfo::pv[0] as space-separated UCI moves (best root move, then
 * Node::get_principal_variation, node.cpp:1111-1121), centipawns[0] (value_to_centipawn of bestMoveQ, evalinfo.cpp:103-112) and
 * movesToMate[0] (non-zero when the position behind the best move is proven; centipawns is 0 then) as update_eval_info leaves them
 * (evalinfo.cpp:195-260).  Returns the number of moves in the line, -1 on error (buffer too small). */
int mi_search_pv(mi_search* sp, int tree, char* uci_line, int cap, int* centipawns, int* moves_to_mate);
/* line idx (0-based) of a Multi_PV output with the UCI option set to multipv (update_eval_info, evalinfo.cpp:195-260; sort_eval_lists
 * :184-193): line 0 = mi_search_pv; line idx >= 1 starts with the root move of rank idx in the MCTS policy (descending, compared as float;
 * equal entries in root order -- the reference's std::sort leaves their order open).  best_move_q = EvalInfo::bestMoveQ[idx].  Returns
 * the number of moves in the line, 0 when idx >= min(multipv, expanded root children), -1 on error. */
int mi_search_pv_multi(mi_search* sp, int tree, int idx, int multipv, char* uci_line, int cap, int* centipawns, int* moves_to_mate, float* best_move_q);
/* One tree, many collectors: the reference runs `Threads` SearchThreads on ONE tree (engine/src/uci/crazyara.cpp:555-561,734;
 * searchthread.cpp:403-416; per-node mutex node.h:100) -- the case of a single UCI `go`.  k >= 1 gives every tree k collectors in
 * EVERY lane: a lane's batch is the concatenation of its collectors' mini-batches (batch / (k * trees) leaves each), collected in
 * parallel by the threads of mi_search_run under per-node locks, virtual loss keeping the descents apart; results are applied and
 * the next leaves collected while the other lanes' batches are on the GPU.  k = 0 (default): one collector per tree, each tree in
 * one lane, no locks (many trees fill the batches instead).  With more than one collector per tree the trees are no longer a
 * deterministic function of the seed (thread timing), exactly as in the reference.  Call between runs. */
int mi_search_set_shared_collectors(mi_search* sp, int k);
/* Throughput setting of the many-trees mode (self-play): cap > 0 = the trees of a lane that are still searching share the whole batch
 * (each batch / running-trees slots, at most cap, never more than the tree still needs); 0 = the reference's fixed per-tree quota. */
int mi_search_set_adaptive_quota(mi_search* sp, int cap);
/* Stored leaf states -- the reference's MCTS_STORE_STATES build option (engine/src/node.h:111,530, searchthread.cpp:198-213): every new
 * node keeps its position, so that an expansion below it copies that state and plays ONE move instead of cloning the root and replaying
 * the whole path.  budget = how many nodes of a tree may hold a state (~0.5 KB each); 0 = off (the default, as in the reference's
 * default build: the replay is 4.5 % of the host's search time on one tree of 25,600 simulations, and with states that search measured
 * 8 - 13 % SLOWER -- a stored state is a position with its history vectors).  The trees are the same bit for bit either way.  Between runs. */
int mi_search_set_state_budget(mi_search* sp, unsigned budget);
/* the whole tree as a flat word list, for inspection and the parity tests (the reference's counterpart: MCTSAgent::export_search_tree,
 * mctsagent.cpp:420-448): depth-first preorder over the expanded children, one record per node that was selected at least once:
 * [n_expanded, visit_sum, real_visits, free_visits, node_type, end_in_ply, terminal, float bits of value], then per expanded child
 * [move, visits, virtual-loss counter, float bits of Q, float bits of prior, state: 0 no node / 1 evaluated, never selected /
 * 2 its record follows].  Returns the number of 32-bit words written, -1 on error (buffer too small). */
long mi_search_tree_dump(mi_search* sp, int tree, uint32_t* out, long cap);
/* Diagnostic of the HIP lanes (process started with CRA_LANE_RECORD=1; otherwise returns 0): every batch the pool's lanes evaluated
 * since the last call is sent through its lane again, alone on the device, and compared bit for bit with what the host found in the
 * lane's result buffers when the batch was reported complete (SearchThread reads them right after predict() returns,
 * searchthread.cpp:403-416) and again when the next batch went out.  Returns the number of differing 32-bit words (0 = every result
 * the searches consumed is reproducible), -1 on error; a text report (NUL-terminated, truncated to cap) goes to `report` if given.
 * Between runs only. */
long mi_search_debug_replay(mi_search* sp, char* report, long cap);
/* active = 0: the tree sits out the following mi_search_run calls and keeps its state (the player that is not to move in an
 * arena game, generate_arena_game, selfplay.cpp:267-308); trees start active */
int mi_search_set_active(mi_search* sp, int tree, int active);
/* start a new game on an existing tree slot: the tree restarts from this position (clean_up / clear_game_history, selfplay.cpp:305-309) */
int mi_search_reset_position(mi_search* sp, int tree, const char* fen, int is_chess960, const char* variant);   /* argmax of Node::get_mcts_policy, node.cpp:1070-1109 */

/* ------------------------------------------------------------------------------------------------------------------
 * Training-sample exporter of the self-play loop: TrainDataExporter (engine/src/rl/traindataexporter.{h,cpp}).  One zarr
 * (format 2) group with the arrays x int16 [N][C][8][8] (un-normalised planes), y_value int16 [N], y_policy float32 [N][NB_LABELS]
 * (classic label index, mirrored for Black), y_best_move_q float32 [N], plys_to_end int16 [N], phase_vector int16 [N],
 * start_indices int32 [N]; N = number_chunks * chunk_size, chunked along N, raw little-endian chunks as z5's default writes them
 * (traindataexporter.cpp:262-283).  mode / version fix the plane layout and the label set (the reference's build flavour).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_traindata mi_traindata;
mi_traindata* mi_traindata_create(const char* path, int mode, int version_major, int version_minor, unsigned number_chunks, unsigned chunk_size);
void mi_traindata_destroy(mi_traindata* t);
/* numPhases / gamePhaseDefinition of TrainDataExporter's constructor (traindataexporter.cpp:136-156; 0 = lichess, 1 = movecount): what
 * save_cur_phase (:91-103) writes into phase_vector for every sample saved through mi_traindata_save_sample / mi_search_save_sample.
 * Default: 1 phase, lichess definition -- the native self-play loop's default. */
int mi_traindata_set_phases(mi_traindata* t, int num_phases, int game_phase_definition);
int mi_traindata_new_game(mi_traindata* t);                                     /* new_game(), :167-171 */
/* save_sample(pos, evalInfo), :33-47: moves = EvalInfo::legalMoves, policy = policyProbSmall (entries beyond n_policy count as 0) */
int mi_traindata_save_sample(mi_traindata* t, const mi_pos* pos, const uint32_t* moves, int n_moves, const double* policy, int n_policy,
                             float best_move_q);
/* the same taken from a searched tree of a pool: root position, its moves, Node::get_mcts_policy, EvalInfo::bestMoveQ */
int mi_search_save_sample(mi_search* sp, int tree, mi_traindata* t);
/* export_game_samples(result), :110-134; result: 0 DRAWN, 1 WHITE_WIN, 2 BLACK_WIN (enum Result, state.h); *written = samples stored */
int mi_traindata_export_game_samples(mi_traindata* t, int result, unsigned* written);
int mi_traindata_info(const mi_traindata* t, unsigned* number_samples, unsigned* start_index, unsigned* game_index, int* nb_labels, int* channels,
                      int* is_full);

/* ------------------------------------------------------------------------------------------------------------------
 * Self-play / arena game loops (engine/src/rl/selfplay.cpp: generate_game :192-265, generate_arena_game :267-308, go_arena :387-424,
 * init_starting_state_from_raw_policy :426-452; agents/agent.cpp set_best_move :38-55; rl/gamepgn.cpp :28-56), native: G games run
 * concurrently on the trees of one pool (arena: of two pools), one pool run searches the next move of all of them.  The pool(s) must
 * be empty: the loop adds one tree per concurrent game.  Random draws come from one seeded generator per game.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct mi_selfplay mi_selfplay;
typedef struct mi_selfplay_settings {
    unsigned simulations, nodes;          /* budget per move (one of them) */
    float node_random_factor;             /* RLSettings::nodeRandomFactor */
    float mean_init_ply;                  /* PlaySettings::meanInitPly: opening plies sampled from the raw policy */
    int max_init_ply;
    float raw_policy_prob_temperature;    /* RLSettings::rawPolicyProbabilityTemperature */
    float init_temperature;               /* PlaySettings::initTemperature / temperatureMoves / temperatureDecayFactor / quantileClipping */
    int temperature_moves;
    float temperature_decay;
    float quantile_clipping;
    float resign_probability;             /* RLSettings::resignProbability / resignThreshold */
    float resign_threshold;
    int reuse_tree;                       /* RLSettings::reuseTreeForSelpay */
    int max_plies;                        /* safety net: adjudicated as a draw */
    unsigned long long seed;
    /* quick searches (SelfPlay::is_quick_search, rl/selfplay.cpp:154-159,213-221): with this probability (< 0.01 = never) a move is
     * searched with quick_search_nodes nodes, its own Q-value weight and Dirichlet epsilon; its position is not exported */
    float quick_search_probability;       /* RLSettings::quickSearchProbability   (Centi_Quick_Probability / 100) */
    unsigned quick_search_nodes;          /* RLSettings::quickSearchNodes         (Quick_Nodes) */
    float quick_search_q_value_weight;    /* RLSettings::quickSearchQValueWeight  (Centi_Quick_Q_Value_Weight / 100) */
    float quick_dirichlet_epsilon;        /* RLSettings::quickDirichletEpsilon    (Centi_Quick_Dirichlet_Epsilon / 100) */
    float low_policy_clip_threshold;      /* RLSettings::lowPolicyClipThreshold: sharpen_distribution on the exported policy (0 = off) */
    int num_phases;                       /* MCTSAgent::get_num_phases: > 1 = one exporter per game phase (mi_selfplay_set_phase_exporter) */
    int game_phase_definition;            /* SearchSettings::gamePhaseDefinition: 0 lichess (1 or 3 phases), 1 movecount */
} mi_selfplay_settings;
typedef struct mi_selfplay_stats {
    unsigned long long moves, nodes, nn_evals, kept_subtrees, restarts, samples;
    double seconds;
    int wins, draws, losses;              /* arena: seen from the contender (pool A) */
    int reserved;
    double run_seconds, move_seconds;     /* of `seconds`: inside the pool's searches / inside the (parallel) move step of the games */
    unsigned long long quick_searches;    /* moves searched in quick mode (not exported) */
    unsigned long long samples_dropped;   /* searched positions that found the export file full (rl/selfplay.cpp:224) */
} mi_selfplay_stats;
void mi_selfplay_default_settings(mi_selfplay_settings* s);
/* pool_b == NULL: self-play on pool_a (exporter: every searched position becomes a training sample, written game by game; may be NULL);
 * pool_b != NULL: arena between pool_a (contender) and pool_b, colours alternating per pair of games (exporter must be NULL). */
mi_selfplay* mi_selfplay_create(mi_search* pool_a, mi_search* pool_b, const mi_selfplay_settings* s, int concurrent, const char* variant,
                                int is_chess960, mi_traindata* exporter);
void mi_selfplay_destroy(mi_selfplay* sp);
/* start positions, '\n'-separated FENs ("" line = the variant's start position): game i (arena: pair i) uses entry i mod count */
int mi_selfplay_set_start_fens(mi_selfplay* sp, const char* fens);
/* RLSettings::epdFilePath (UCI option EPD_File_Path, optionsuci.cpp:206; load_random_fen, rl/selfplay.cpp:58-80,201,396): every game
 * (arena: every colour-swapped pair of games) starts from a random line of the EPD file -- one FEN per line, a trailing ';' dropped --
 * drawn with the game's own seeded generator.  "" or "<empty>" = no file (back to mi_selfplay_set_start_fens / the start position);
 * an unreadable or empty file is an error.  Overrides mi_selfplay_set_start_fens. */
int mi_selfplay_set_epd_file(mi_selfplay* sp, const char* path);
/* self-play with num_phases > 1: the exporter of game phase `phase` >= 1 (phase 0 is mi_selfplay_create's); every sample goes to the
 * exporter of its position's phase (rl/selfplay.cpp:232-238, Board::get_phase, board.cpp:540-587) */
int mi_selfplay_set_phase_exporter(mi_selfplay* sp, int phase, mi_traindata* exporter);
/* n_games > 0: plays until n_games are finished in total (SelfPlay::go(N): every game is played out; positions beyond the export
 * file's capacity are searched and dropped, mi_selfplay_stats::samples_dropped).  n_games == 0 (self-play with an exporter):
 * SelfPlay::go(0), rl/selfplay.cpp:374-377 -- games are started until the export file is full, running games are played out.
 * Returns the total of finished games, -1 on error */
int mi_selfplay_play(mi_selfplay* sp, int n_games, int threads);
/* finished game `index`: result +1 / 0 / -1 for White, plies from the opening book, whether the contender had White (arena), and as
 * text "start FEN\ntermination\nSAN moves separated by \t\nUCI moves separated by \t\n".  Returns the text's length (call with cap 0
 * to size the buffer), -1 on error. */
long mi_selfplay_game(mi_selfplay* sp, int index, int* result, int* book_plies, int* contender_white, char* text, long cap);
int mi_selfplay_get_stats(mi_selfplay* sp, mi_selfplay_stats* out);
/* The sampling helpers of Agent::set_best_move as the loops use them, in place on p[n] (for tests against the reference's functions):
 * apply_temperature (blazeutil.h:77-87), get_quantile (:188-212), apply_quantile_clipping (agent.cpp:121-130). */
void mi_policy_apply_temperature(double* p, int n, double temperature);
double mi_policy_get_quantile(const double* p, int n, double quantile);
void mi_policy_apply_quantile_clipping(double* p, int n, double quantile);
void mi_policy_sharpen_distribution(double* p, int n, double thresh);            /* sharpen_distribution, util/blazeutil.h:94-105 */

#ifdef __cplusplus
}
#endif
#endif /* CRAZYARA_HIP_H */
