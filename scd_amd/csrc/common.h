// Internal helpers shared by the HIP translation units of libscd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <math.h>
#include "../../include/scd_hip.h"

#ifdef __cplusplus
#include <utility>
#include <vector>
#endif

struct scd_ctx {
    int device;
    int n_cu;
    void* scratch;          // 256 KB + 64 B of device memory, zero at creation: per-centre partials + ticket of scd_kmeans_finalize
    // set by scd_kmeans_finalize when it also wrote the E-step operands of C_out into an E-step workspace; consumed (and
    // cleared) by the next scd_kmeans_estep on the same handle: a match skips the centre-prep launch
    const void* prep_C;
    const void* prep_ws;
    int prep_k, prep_d;
    int prep_ok;            // scd_kmeans_estep_hint(SCD_ESTEP_CENTRES_FROM_FINALIZE): the caller vouches that the next E-step's centres are prep_C's, unmodified
    // scd_kmeans_timing: while enabled, the streaming E-step kernel launches of scd_kmeans_estep (hence of scd_kmeans_lloyd_step)
    // are bracketed by HIP events on their launch stream
    bool km_timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> km_ev;
    int estep_few;          // scd_kmeans_estep_hint: the next E-step re-evaluates its (few) flagged rows in the filter kernel's tail
    // scd_kmeans_lloyd_run: pinned, device-mapped host ring (2 x 8 doubles) finalize_kernel publishes an iteration's statistics in
    // ({inertia l, inertia u, shift, refined, changed, -, -, sequence number}), its device address, the last sequence number used
    double* run_host = nullptr;
    double* run_dev = nullptr;
    double run_seq = 0.0;
};
#define SCD_SCRATCH_BYTES (262144 + 64)

void scd_set_error(const char* fmt, ...);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (current device, kernel): the attribute is per device, so a
// process-wide `static bool` would leave a second device without it (api.cpp; mutex-protected)
int scd_set_max_lds(const void* fn, int bytes);
// the handle's device must be the current one: every launch below goes to the current device
int scd_check_device(const struct scd_ctx* h, const char* who);

#define SCD_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) {                                          \
            scd_set_error(__VA_ARGS__);                         \
            return SCD_EINVAL;                                  \
        }                                                       \
    } while (0)

#define SCD_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            scd_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SCD_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

#define SCD_LAUNCH_CHECK() SCD_HIP(hipGetLastError())
// first statement of every entry point that launches on the handle's device
#define SCD_DEVICE_ENTRY(h, who)                                \
    do {                                                        \
        SCD_REQUIRE((h) != nullptr, who ": null handle");       \
        const int rc_dev_ = scd_check_device((h), who);         \
        if (rc_dev_) return rc_dev_;                            \
    } while (0)

// Environment switches that can change RESULTS (timing ablations: kernels with pieces removed) or that select kernels kept only for
// A/B measurements exist in builds with -DSCD_ABLATE (`python -m scd_amd.build --ablate` -> lib/libscd_hip_ablate.so, loaded through
// SCD_HIP_LIB).  The default library does not read these variables and does not contain the code behind them.
#ifdef SCD_ABLATE
#include <stdlib.h>
#define SCD_ABLATE_ENV(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define SCD_ABLATE_ENV(name, dflt) (dflt)
#endif

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline size_t scd_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline int64_t scd_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
#include <type_traits>
// compile-time loop: f(std::integral_constant<int, S>) for S = B .. E-1 (asm immediates need constant expressions)
template <int B, int E, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}
__host__ __device__ __forceinline__ long long scd_cdiv_dev(long long a, long long b) { return (a + b - 1) / b; }
// ---- wavefront (64 lanes) reductions ----
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif
