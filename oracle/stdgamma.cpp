// ORACLE helper (test infrastructure only).  The reference's Dirichlet noise is "one std::gamma_distribution<float>(alpha, 1)
// per entry, drawn from a std::default_random_engine" (engine/src/util/blazeutil.h:113-124, util/randomgen.h:35): the
// arithmetic lives in the C++ standard library, not in the reference, so the oracle calls the same library through this shim
// (libstdc++: default_random_engine = minstd_rand0) instead of restating it in Python.  Built by oracle/build_oracle.py.
#include <random>

extern "C" {
void* stdgamma_new(unsigned seed) { return new std::minstd_rand0(seed); }
void stdgamma_free(void* e) { delete static_cast<std::minstd_rand0*>(e); }
// out[i] = gamma_distribution<float>(alpha, 1)(engine), a fresh distribution object per entry as the reference constructs it
void stdgamma_draw(void* e, float alpha, int n, float* out) {
    std::minstd_rand0& g = *static_cast<std::minstd_rand0*>(e);
    for (int i = 0; i < n; ++i) {
        std::gamma_distribution<float> distribution(alpha, 1.0f);
        out[i] = distribution(g);
    }
}
}
