// ORACLE BUILD SUPPORT: the reference's node.cpp compiled as it is, plus one hook into its translation unit.
// util/randomgen.h:35-36 gives every translation unit that includes blazeutil.h its own `static std::default_random_engine
// generator(r())`, seeded from std::random_device; Node::apply_dirichlet_noise_to_prior_policy (node.cpp:950-954) draws from the
// one of node.cpp.  Re-seeding it is only possible from inside that translation unit, hence this wrapper.
#include "node.cpp"

void refshim_seed_node_generator(unsigned seed) { generator.seed(seed); }
