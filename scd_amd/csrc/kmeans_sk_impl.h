// kmeans_sk_impl.h - included at the end of kmeans.hip (it uses that file's static kernels and helpers; not a stand-alone header).
//
// The `--cluster KM` path of the reference (`KMeans(n_clusters, random_state=0).fit(u_feats).labels_`, /root/reference/main_unsup.py:362,
// main_ptsup.py:381; the default clustering of scripts/evaluate_unsupervised.sh) on the machinery the SSKM path got in round 3:
//   scd_kpp_greedy_lockstep   scikit-learn's greedy k-means++ (`_k_init` of the scikit-learn the reference vendors,
//                             local_utils/k_means_constrained/sklearn_import/cluster/k_means_.py:33-132 = `_kmeans_plusplus` of 1.0.2 /
//                             1.7.2) for the n_init starts of a fit in lock-step, the rounds in C;
//   scd_kmeans_lloyd_run_sk   `_kmeans_single_lloyd` (strict / tolerance stop, final E-step) with the host one iteration behind.
//
// Greedy seeding, one round (centre t of every start r < R): L = 2 + int(ln k) candidates per start by searchsorted on the cumulative
// closest distances; the candidate with the smallest new potential sum_i min(d2[r][i], ||x_i - x_cand||^2) is kept.  With the exact fp16
// copy of X the M = R * L candidates of a round go through the MFMA lower-bound filter of the lock-step seeding (muf_filter_kernel, 64
// candidates per pass over X: four blocks of one XCD hold 16 candidates each and walk the same row tiles together): for most rows the bound proves min(d2, dist) = d2, the (row, candidate) pairs it cannot rule out get the
// exact float32(float64 sum) distance, and a candidate's potential is sum(d2) + sum over its listed pairs of (min(d2, dist) - d2): the
// same float32 values summed in float64 as a dense evaluation would.  The winner's listed values are then written into d2.  Without the
// copy (float32 features that do not survive the fp16 round trip) and in the first rounds (a candidate improves most rows) the
// candidates are evaluated densely by the tile kernel of the lock-step seeding (minupd_tile_kernel).

// candidates: idx_out[r * L + l] = searchsorted(cumsum_f64(d2[r]), u[r * L + l] * float32(sum d2[r])) clipped to n - 1; grid (L, R)
__global__ void __launch_bounds__(1024) kg_search_kernel(const float* __restrict__ d2_all, long long n, long long ld, const double* __restrict__ u,
                                                         const double* __restrict__ bsum_all, int nb, int L, long long* __restrict__ idx_out) {
    __shared__ double sh[32];
    __shared__ long long best;
    __shared__ int owner;
    __shared__ double owner_pre;
    const int r = blockIdx.y, slot = r * L + blockIdx.x;
    const float* d2 = d2_all + (size_t)r * ld;
    const double* bsum = bsum_all + (size_t)r * nb;
    __shared__ double s_bs[KPP_STAGE];                             // thread 0 walks the tile sums twice: from LDS, not as dependent L2 reads
    if (nb <= KPP_STAGE) {
        for (int b = threadIdx.x; b < nb; b += 1024) s_bs[b] = bsum[b];
        __syncthreads();
        bsum = s_bs;
    }
    if (threadIdx.x == 0) {
        double pot = 0.0;
        for (int b = 0; b < nb; ++b) pot += bsum[b];
        const double rv = u[slot] * (double)(float)pot;
        double run = 0.0;
        int ow = -1;
        double opre = 0.0;
        for (int b = 0; b < nb; ++b) {
            const double nxt = run + bsum[b];
            if (ow < 0 && nxt >= rv) { ow = b; opre = run; }
            run = nxt;
        }
        owner = ow;
        owner_pre = opre;
        best = 0x7fffffffffffffffll;
        sh[31] = rv;
    }
    __syncthreads();
    const double rv = sh[31];
    if (owner < 0) {
        if (threadIdx.x == 0) idx_out[slot] = n - 1;
        return;
    }
    double pre = owner_pre;
    for (int b = owner; b < nb; ++b) {
        const long long i0 = (long long)b * KPP_TILE + threadIdx.x * 4;
        float pv[4];
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pv[q] = (i0 + q < n) ? d2[i0 + q] : 0.f;
            s += (double)pv[q];
        }
        double tot;
        double run = pre + block_scan_excl_1024(s, sh, &tot);
        long long found = 0x7fffffffffffffffll;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            run += (double)pv[q];
            if (found == 0x7fffffffffffffffll && i0 + q < n && run >= rv) found = i0 + q;
        }
        if (found != 0x7fffffffffffffffll) atomicMin((unsigned long long*)&best, (unsigned long long)found);
        __syncthreads();
        if (best != 0x7fffffffffffffffll) break;
        pre += tot;
    }
    if (threadIdx.x == 0) idx_out[slot] = (best == 0x7fffffffffffffffll) ? n - 1 : best;
}

// block m < Mp (Mp = M rounded up to 64): Cn[m] = X[cand[m]] (float32), c16[m] = its fp16 image (zero rows beyond M, zero columns
// beyond d), info[m] = {||c16||^2, ||c - c16||}; block 0 also clears the potentials of the round
__global__ void __launch_bounds__(256) kg_prep_kernel(const float* __restrict__ X, const long long* __restrict__ cand, int M, int d, int dp,
                                                      float* __restrict__ Cn, half_t* __restrict__ c16, double* __restrict__ info,
                                                      double* __restrict__ potd) {
    __shared__ double red[4][2];
    const int m = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (m == 0)
        for (int i = threadIdx.x; i < M; i += 256) potd[i] = 0.0;
    double s2 = 0.0, e2 = 0.0;
    const long long src = m < M ? cand[m] : 0;
    for (int j = threadIdx.x; j < dp; j += 256) {
        const float c = (m < M && j < d) ? X[src * d + j] : 0.f;
        if (m < M && j < d) Cn[(size_t)m * d + j] = c;
        if (c16) {
            const half_t hc = (half_t)c;
            c16[(size_t)m * dp + j] = hc;
            const double hv = (double)(float)hc, e = (double)c - hv;
            s2 = fma(hv, hv, s2);
            e2 = fma(e, e, e2);
        }
    }
    if (!c16) return;
    s2 = wave_sum_f64(s2);
    e2 = wave_sum_f64(e2);
    if (lane == 0) { red[wave][0] = s2; red[wave][1] = e2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        info[m * 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        info[m * 2 + 1] = sqrt((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    }
}

// the listed (row, candidate) pairs of batch blockIdx.y: v = min(d2[m / L][row], float32(sum_j (x_j - c_j)^2)) (float64 accumulation, x
// from the exact fp16 copy: muf_exact_kernel's value), kept in vals[] for the winner's update; potd[m] += v - d2 (float64, <= 0).
// EXCEPTION to the library's fixed-order-reduction rule (the only one): the per-candidate differences are added with float64 atomics
// (LDS, then global), so the ORDER of the additions - hence the last bits of potd - may differ from run to run, and the filter path ranks
// the candidates by sum(delta) where the dense path ranks them by fixed-order full sums.  The values themselves (vals[], d2) are exact
// and order-free; what can differ is a pick between two candidates whose potentials agree to ~1e-16 relative (2 x 10^-16 x |pot| against
// gaps of the order of pot / n), which the reference's own float32 potentials cannot resolve either.  The picks of every golden seeding
// (tests/golden/kmeans_sklearn.npz, 42 seedings incl. the reference-held `_k_init`) are reproduced on both paths.
__global__ void __launch_bounds__(256) kg_exact_kernel(const half_t* __restrict__ X16, const float* __restrict__ Cn, int d, int L, int M,
                                                       const unsigned* __restrict__ counts_all, const unsigned long long* __restrict__ list_all,
                                                       float* __restrict__ vals_all, long long cap, int g, const float* __restrict__ d2,
                                                       long long ld, double* __restrict__ potd) {
    __shared__ double lpot[256];
    const int lane = threadIdx.x & 63, sub = lane >> 4, l16 = lane & 15;
    const size_t region = (size_t)blockIdx.y * g + blockIdx.x;
    const unsigned cnt = counts_all[region];
    const unsigned long long* mine = list_all + region * cap;
    float* myv = vals_all + region * cap;
    lpot[threadIdx.x] = 0.0;
    __syncthreads();
    for (unsigned p0 = (threadIdx.x >> 6) * 4; p0 < cnt; p0 += 16) {
        const unsigned p = p0 + sub;
        const bool on = p < cnt;
        const unsigned long long e = mine[on ? p : p0];
        const long long row = (long long)(e >> 8);
        const int m = (int)(e & 255);
        const half_t* x = X16 + row * d;
        const float* c = Cn + (size_t)m * d;
        const float old = d2[(size_t)(m / L) * ld + row];
        double s = 0.0;
        for (int j0 = l16 * 8; j0 < d; j0 += 512) {
            half8 xv[4];
            float4 cv[4][2];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 128 * u;
                if (j < d) {
                    xv[u] = *(const half8*)(x + j);
                    cv[u][0] = *(const float4*)(c + j);
                    cv[u][1] = *(const float4*)(c + j + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 128 * u;
                if (j < d) {
                    const float cc[8] = {cv[u][0].x, cv[u][0].y, cv[u][0].z, cv[u][0].w, cv[u][1].x, cv[u][1].y, cv[u][1].z, cv[u][1].w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const double t = (double)(float)xv[u][q] - (double)cc[q];
                        s = fma(t, t, s);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (on && l16 == 0) {
            const float v = fminf(old, (float)s);
            myv[p] = v;
            if (v < old) atomicAdd(&lpot[m], (double)v - (double)old);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < M && lpot[threadIdx.x] != 0.0) atomicAdd(&potd[threadIdx.x], lpot[threadIdx.x]);
}

// best[r] = the first candidate of start r with the smallest potential (np.argmin / the strict `<` of `_k_init`); NaN never wins
__device__ __forceinline__ int kg_best_of(const double* __restrict__ potd, int r, int L) {
    int b = 0;
    double bv = potd[r * L];
    for (int l = 1; l < L; ++l) {
        const double v = potd[r * L + l];
        if (v < bv || (bv != bv && v == v)) { bv = v; b = l; }
    }
    return b;
}
// filter path: the winner's listed values into d2; block (0, 0) also records the round's picks and centres
__global__ void __launch_bounds__(256) kg_apply_kernel(const unsigned* __restrict__ counts_all, const unsigned long long* __restrict__ list_all,
                                                       const float* __restrict__ vals_all, long long cap, int g, const double* __restrict__ potd,
                                                       int R, int L, float* __restrict__ d2, long long ld, const long long* __restrict__ cand,
                                                       const float* __restrict__ Cn, int d, float* __restrict__ C_slot, long long ldc,
                                                       long long* __restrict__ picks_t) {
    __shared__ int win[64];
    if ((int)threadIdx.x < R) win[threadIdx.x] = threadIdx.x * L + kg_best_of(potd, threadIdx.x, L);
    __syncthreads();
    const size_t region = (size_t)blockIdx.y * g + blockIdx.x;
    const unsigned cnt = counts_all[region];
    const unsigned long long* mine = list_all + region * cap;
    const float* myv = vals_all + region * cap;
    for (unsigned p = threadIdx.x; p < cnt; p += 256) {
        const unsigned long long e = mine[p];
        const int m = (int)(e & 255), r = m / L;
        if (win[r] == m) d2[(size_t)r * ld + (long long)(e >> 8)] = myv[p];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        for (int r = 0; r < R; ++r) {
            const int m = win[r];
            if (threadIdx.x == 0) picks_t[r] = cand[m];
            for (int j = threadIdx.x; j < d; j += 256) C_slot[(size_t)r * ldc + j] = Cn[(size_t)m * d + j];
        }
    }
}
// dense path: potd[r * L + l] holds the whole potential of trial l (sum of tmp[l][r]); d2[r] = tmp[winner][r]; block (0, 0) records
__global__ void __launch_bounds__(256) kg_select_kernel(const float* __restrict__ tmp, long long n, long long ld, const double* __restrict__ potd,
                                                        int R, int L, float* __restrict__ d2, const long long* __restrict__ cand,
                                                        const float* __restrict__ Cn, int d, float* __restrict__ C_slot, long long ldc,
                                                        long long* __restrict__ picks_t) {
    const int r = blockIdx.y;
    const int b = kg_best_of(potd, r, L);
    const float* src = tmp + ((size_t)b * R + r) * ld;
    float* dst = d2 + (size_t)r * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0) {
        const int m = r * L + b;
        if (threadIdx.x == 0) picks_t[r] = cand[m];
        for (int j = threadIdx.x; j < d; j += 256) C_slot[(size_t)r * ldc + j] = Cn[(size_t)m * d + j];
    }
}
// out[r * stride] = sum_b bsum[r][b], tiles in index order (thread r < R)
__global__ void kg_tiles_pot_kernel(const double* __restrict__ bsum, int nb, int R, double* __restrict__ out, int stride) {
    const int r = threadIdx.x;
    if (r >= R) return;
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += bsum[(size_t)r * nb + b];
    out[(size_t)r * stride] = t;
}
// potd[r * L + l] = sum_b bsum2[l * R + r][b], tiles in index order: all trials of a round in one launch
__global__ void __launch_bounds__(256) kg_tiles_pot_all_kernel(const double* __restrict__ bsum2, int nb, int R, int L, double* __restrict__ potd) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= R * L) return;
    const int l = m / R, r = m % R;
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += bsum2[(size_t)m * nb + b];
    potd[(size_t)r * L + l] = t;
}
__global__ void __launch_bounds__(256) kg_fill_kernel(float* __restrict__ p, long long n_elems, float v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (long long)gridDim.x * 256) p[i] = v;
}

struct KgLayout {
    long long ld, cap;
    int g, dp, nbat, Mp, nb;
    size_t o_d2, o_bsum, o_bsum2, o_cand, o_potd, o_cn, o_c16, o_info, o_rn2, o_counts, o_list, o_vals, o_tmp, total;
};
static KgLayout kg_layout(int64_t n, int d, int R, int L, bool filt) {
    KgLayout y;
    const int M = R * L;
    y.ld = (n + 63) / 64 * 64;
    y.dp = muf_dp(d);
    y.g = muf_grid(y.dp);
    y.cap = scd_cdiv(scd_cdiv(n, 16), y.g) * 1024;       // a block's row tiles (its four waves': ceil(tiles / g) each) x 16 rows x 16 candidates
    y.nbat = (M + 63) / 64;                               // passes over X: 64 candidates each (four groups of 16, one per wave)
    y.Mp = y.nbat * 64;
    y.nb = (int)scd_cdiv(n, KPP_TILE);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += scd_align(bytes); return at; };
    y.o_d2 = take(4 * (size_t)R * y.ld);
    y.o_bsum = take(8 * (size_t)R * y.nb);
    y.o_bsum2 = take(8 * (size_t)R * L * y.nb);            // per-tile sums of the L trials' distance vectors (dense path)
    y.o_cand = take(8 * (size_t)y.Mp);
    y.o_potd = take(8 * (size_t)y.Mp);
    y.o_cn = take(4 * (size_t)y.Mp * d);
    y.o_c16 = take(2 * (size_t)y.Mp * y.dp);
    y.o_info = take(16 * (size_t)y.Mp);
    y.o_rn2 = take(4 * (size_t)n);
    y.o_counts = take(4 * (size_t)y.nbat * y.g);
    y.o_list = take(filt ? 8 * (size_t)y.nbat * y.g * y.cap : 0);
    y.o_vals = take(filt ? 4 * (size_t)y.nbat * y.g * y.cap : 0);
    y.o_tmp = take(4 * (size_t)L * R * y.ld);          // dense path (first rounds, or no exact fp16 copy)
    y.total = o + 256;
    return y;
}
static bool kg_can_filter(int64_t n, int d, int R, int L) {
    const int dp = muf_dp(d);
    return R * L <= 256 && R <= 64 && d % 32 == 0 && (dp == 128 || dp == 256 || dp == 384 || dp == 512 || dp == 768) && n < (1ll << 40);
}
extern "C" size_t scd_kpp_greedy_ws_bytes(int64_t n, int d, int R, int L) { return kg_layout(n, d, R, L, kg_can_filter(n, d, R, L)).total; }

extern "C" int scd_kpp_greedy_lockstep(scd_handle h, const float* X, const void* X16, int64_t n, int d, int R, int L, int k,
                                       const int64_t* first, const double* u_dev, float* C_buf, int64_t* picks_out, void* ws,
                                       size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kpp_greedy_lockstep");
    SCD_REQUIRE(X && first && C_buf && picks_out && ws && n > 0 && d > 0 && R > 0 && R <= 64 && L > 0 && L <= 64 && k >= 1 && (k == 1 || u_dev),
                "scd_kpp_greedy_lockstep: bad arguments");
    const bool can = kg_can_filter(n, d, R, L);
    SCD_REQUIRE(ws_bytes >= kg_layout(n, d, R, L, can).total, "scd_kpp_greedy_lockstep: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const KgLayout y = kg_layout(n, d, R, L, can);
    char* w = (char*)ws;
    float* d2 = (float*)(w + y.o_d2);
    double* bsum = (double*)(w + y.o_bsum);
    double* bsum2 = (double*)(w + y.o_bsum2);
    long long* cand = (long long*)(w + y.o_cand);
    double* potd = (double*)(w + y.o_potd);
    float* Cn = (float*)(w + y.o_cn);
    half_t* c16 = (half_t*)(w + y.o_c16);
    double* info = (double*)(w + y.o_info);
    float* rn2 = (float*)(w + y.o_rn2);
    unsigned* counts = (unsigned*)(w + y.o_counts);
    unsigned long long* list = (unsigned long long*)(w + y.o_list);
    float* vals = (float*)(w + y.o_vals);
    float* tmp = (float*)(w + y.o_tmp);
    const int M = R * L;
    const long long ldc = (long long)k * d;
    static const int from_env = getenv("SCD_KM_FILTER_FROM") ? atoi(getenv("SCD_KM_FILTER_FROM")) : 4;   // centres before the filter pays
    const bool filt = can && X16 != nullptr && from_env >= 0;
    // centre 0 of every start: the row the host's RandomState drew; d2 = its distances
    kpp_fetch_rows_kernel<<<R, 256, 0, st>>>(X, (const long long*)first, d, C_buf, ldc);
    SCD_HIP(hipMemcpyAsync(picks_out, first, 8 * (size_t)R, hipMemcpyDeviceToDevice, st));
    kg_fill_kernel<<<1024, 256, 0, st>>>(d2, (long long)R * y.ld, __builtin_inff());
    minupd_all(X, C_buf, n, d, R, d2, y.ld, ldc, st);
    if (filt && k > 1) muf_rown2_kernel<<<(unsigned)scd_cdiv(n, 4), 256, 0, st>>>((const half_t*)X16, n, d, rn2);
    for (int t = 1; t < k; ++t) {
        kpp_tile_sum_multi_kernel<<<dim3(y.nb, R), 1024, 0, st>>>(d2, n, y.ld, bsum, y.nb);
        kg_search_kernel<<<dim3(L, R), 1024, 0, st>>>(d2, n, y.ld, u_dev + (size_t)(t - 1) * M, bsum, y.nb, L, cand);
        const bool use_filter = filt && t >= from_env;
        kg_prep_kernel<<<use_filter ? y.Mp : M, 256, 0, st>>>(X, cand, M, d, y.dp, Cn, use_filter ? c16 : nullptr, info, potd);
        float* slot = C_buf + (size_t)t * d;
        long long* picks_t = (long long*)picks_out + (size_t)t * R;
        if (use_filter) {
            for (int b = 0; b < y.nbat; ++b) {
                // groups of 16 candidates in this batch of up to 64: every group is a pass over X16, so a single start (L = 6 candidates)
                // gets ONE pass on the whole grid instead of four quarter-grids of which three had nothing to measure (94 -> 33 us per
                // round); the regions [0, g) of counts / list are all written in each form
                const int Mb = M - 64 * b < 64 ? M - 64 * b : 64;
                const int ngrp = Mb <= 16 ? 1 : Mb <= 32 ? 2 : 4;
#define KG_ARGS (const half_t*)X16, rn2, c16 + (size_t)b * 64 * y.dp, info + b * 128, n, d, M, d2, y.ld, counts + (size_t)b * y.g, \
                list + (size_t)b * y.g * y.cap, y.cap, b * 64, L
#define KG_GO(NKS)                                                                                                        \
    if (ngrp == 1) muf_filter_kernel<NKS, 1><<<y.g, 256, 0, st>>>(KG_ARGS);                                               \
    else muf_filter_kernel<NKS, 4><<<dim3(y.g / ngrp, ngrp), 256, 0, st>>>(KG_ARGS)
                switch (y.dp / 32) {
                    case 4: KG_GO(4); break;
                    case 8: KG_GO(8); break;
                    case 12: KG_GO(12); break;
                    case 16: KG_GO(16); break;
                    default: KG_GO(24); break;
                }
#undef KG_GO
#undef KG_ARGS
            }
            kg_exact_kernel<<<dim3(y.g, y.nbat), 256, 0, st>>>((const half_t*)X16, Cn, d, L, M, counts, list, vals, y.cap, y.g, d2, y.ld, potd);
            kg_apply_kernel<<<dim3(y.g, y.nbat), 256, 0, st>>>(counts, list, vals, y.cap, y.g, potd, R, L, d2, y.ld, cand, Cn, d, slot, ldc, picks_t);
        } else {
            // the L trials of the round in ONE distance launch per restart group (blockIdx.y = trial): tmp[l][r] = min(d2[r], distances
            // to trial l's candidate of restart r) - d2 is read, not copied L times (round 6: 4 L + 4 launches per round -> 7)
            minupd_all(X, Cn, n, d, R, tmp, y.ld, (long long)L * d, st, d2, L, d, (long long)R * y.ld);
            // potentials: potd[r * L + l] = float64 sum of tmp[l][r]: per-tile sums over the whole chip (rows l * R + r of tmp), then the
            // tiles in index order (one 1024-thread block per row took 108 us per trial at 95,000 rows)
            kpp_tile_sum_multi_kernel<<<dim3(y.nb, R * L), 1024, 0, st>>>(tmp, n, y.ld, bsum2, y.nb);
            kg_tiles_pot_all_kernel<<<(unsigned)scd_cdiv(M, 256), 256, 0, st>>>(bsum2, y.nb, R, L, potd);
            kg_select_kernel<<<dim3(64, R), 256, 0, st>>>(tmp, n, y.ld, potd, R, L, d2, cand, Cn, d, slot, ldc, picks_t);
        }
    }
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------ one filtered distance update
// scd_kpp_update_filter: d2[r] = min(d2[r], ||x - c_new[r]||^2) for R restarts through the MFMA lower-bound filter of
// scd_kpp_seed_lockstep (muf_filter_kernel + muf_exact_kernel) as a call of its own - the lock-step seeding under a process group
// (scd_amd/kmeans.py kpp_lockstep: three all-gathers sit between a round's draw and its update, so the rounds cannot run behind one
// call) gets the same update as the single-process loop instead of the float32 tile kernel.  c_new [R][d] float32: rows of the GLOBAL
// X (any rank's shard), hence exact in fp16 like the local rows; the bound covers centres that are not.
// block r < 16: c16[r] = fp16(c_new[r]) (zero rows beyond R), info[r] = {||c16_r||^2, ||c_r - c16_r||}
__global__ void __launch_bounds__(256) muf_prep_rows_kernel(const float* __restrict__ Cn, int R, int d, int dp, half_t* __restrict__ c16,
                                                            double* __restrict__ info) {
    __shared__ double red[4][2];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s2 = 0.0, e2 = 0.0;
    for (int j = threadIdx.x; j < dp; j += 256) {
        const float c = (r < R && j < d) ? Cn[(size_t)r * d + j] : 0.f;
        const half_t hc = (half_t)c;
        c16[(size_t)r * dp + j] = hc;
        const double hv = (double)(float)hc, e = (double)c - hv;
        s2 = fma(hv, hv, s2);
        e2 = fma(e, e, e2);
    }
    s2 = wave_sum_f64(s2);
    e2 = wave_sum_f64(e2);
    if (lane == 0) { red[wave][0] = s2; red[wave][1] = e2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        info[r * 2] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        info[r * 2 + 1] = sqrt((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    }
}
static bool muf_shape_ok(int64_t n, int d, int R) {
    const int dp = muf_dp(d);
    return R >= 1 && R <= 16 && d % 32 == 0 && (dp == 128 || dp == 256 || dp == 384 || dp == 512 || dp == 768) && n < (1ll << 40);
}
extern "C" size_t scd_kpp_update_ws_bytes(int64_t n, int d) {
    const int g = muf_grid(muf_dp(d));
    return scd_align(4 * (size_t)n) + scd_align(2 * 16 * (size_t)muf_dp(d)) + 256 + scd_align(4 * 1024) + scd_align(8 * (size_t)muf_cap(n, g) * g) + 256;
}
extern "C" int scd_kpp_update_filter(scd_handle h, const void* X16, int64_t n, int d, int R, const float* c_new, float* d2, int64_t ld,
                                     int first_call, void* ws, size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_kpp_update_filter");
    SCD_REQUIRE(X16 && c_new && d2 && ws && n > 0 && ld >= n, "scd_kpp_update_filter: bad arguments");
    SCD_REQUIRE(muf_shape_ok(n, d, R), "scd_kpp_update_filter: shape not served by the filter (R=%d d=%d): use scd_kmeans_min_update_multi", R, d);
    SCD_REQUIRE(ws_bytes >= scd_kpp_update_ws_bytes(n, d), "scd_kpp_update_filter: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const int dp = muf_dp(d), g = muf_grid(dp);
    const long long cap = muf_cap(n, g);
    char* w = (char*)ws;
    float* rn2 = (float*)w;
    half_t* c16 = (half_t*)(w + scd_align(4 * (size_t)n));
    double* info = (double*)((char*)c16 + scd_align(2 * 16 * (size_t)dp));
    unsigned* counts = (unsigned*)((char*)info + 256);
    unsigned long long* list = (unsigned long long*)((char*)counts + scd_align(4 * 1024));
    if (first_call) muf_rown2_kernel<<<(unsigned)scd_cdiv(n, 4), 256, 0, st>>>((const half_t*)X16, n, d, rn2);   // the table lives in ws between calls
    muf_prep_rows_kernel<<<16, 256, 0, st>>>(c_new, R, d, dp, c16, info);
#define MUF_GO(NKS) muf_filter_kernel<NKS><<<g, 256, 0, st>>>((const half_t*)X16, rn2, c16, info, n, d, R, d2, ld, counts, list, cap)
    switch (dp / 32) {
        case 4: MUF_GO(4); break;
        case 8: MUF_GO(8); break;
        case 12: MUF_GO(12); break;
        case 16: MUF_GO(16); break;
        default: MUF_GO(24); break;
    }
#undef MUF_GO
    muf_exact_kernel<<<g, 256, 0, st>>>((const half_t*)X16, c_new, d, d, counts, list, cap, d2, ld);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------ sklearn's Lloyd loop in C
// `_kmeans_single_lloyd` (scikit-learn, third-party; call site /root/reference/main_unsup.py:362) for rows with an exact fp16 copy:
//     for i < max_iter: E-step, centre update; stop when no label changed (strict; never at i = 0) or sum_k ||dc_k||^2 <= tol;
//     afterwards the E-step of the final centres (the labels `KMeans.labels_` returns).
// Iteration i is scd_kmeans_lloyd_step_delta (shift mode 1) with the host one iteration behind, as in scd_kmeans_lloyd_run; the
// speculative iteration i + 1 that is always in flight when i turns out to be the last one IS the final E-step - its labels are the
// result, its centre update is discarded.  final_C = iteration i's centres.  result_host (HOST double [4]) = {status, n_iter,
// iterations with the incremental M-step, iterations launched}; status 1: an iteration left a cluster empty (sklearn relocates it
// to a far point: `_relocate_empty_clusters_dense`) - nothing valid is returned and the caller runs that start itself.
extern "C" int scd_kmeans_lloyd_run_sk(scd_handle h, const float* X, const void* prep, int64_t n, const void* X16, int d, int k,
                                       int32_t* lab_ring, int32_t* labels_prev, const float* C_start, float* C_ring, double* sums,
                                       int64_t* counts, const double* sumsq4, double* stats_ring, int max_iter, double tol,
                                       int32_t* final_labels, float* final_C, double* result_host, void* ws_e, size_t ws_e_bytes,
                                       void* ws_m, size_t ws_m_bytes, void* stream) {
    SCD_DEVICE_ENTRY(h, "scd_kmeans_lloyd_run_sk");
    SCD_REQUIRE(X && prep && X16 && lab_ring && labels_prev && C_start && C_ring && sums && counts && sumsq4 && stats_ring && final_labels &&
                    final_C && result_host, "scd_kmeans_lloyd_run_sk: null argument");
    SCD_REQUIRE(max_iter >= 1 && n > 0, "scd_kmeans_lloyd_run_sk: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!h->run_host) {
        SCD_HIP(hipHostMalloc((void**)&h->run_host, 2 * 8 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
        for (int i = 0; i < 16; ++i) h->run_host[i] = 0.0;
        SCD_HIP(hipHostGetDevicePointer((void**)&h->run_dev, h->run_host, 0));
    }
    h->prep_C = nullptr;
    h->prep_ok = 0;
    const size_t kd = (size_t)k * d;
    double refined_seen = -1., changed_seen = -1., changed_prev = -1.;
    const double many = (double)(n / 32 > 256 ? n / 32 : 256);
    double seq_of[2] = {0., 0.};
    int delta_steps = 0, launched = 0, stop_it = -1, status = 0;
    // wait for iteration i's statistics: 0 = go on, 1 = i is the last iteration, 2 = empty cluster, < 0 = error
    auto settle = [&](int i) -> int {
        volatile double* host = h->run_host + (i & 1) * 8;
        const double want = seq_of[i & 1];
        long long spins = 0;
        while (host[7] != want) {
            if ((++spins & 0xFFFFF) == 0) {
                const hipError_t e = hipStreamQuery(st);
                if (e != hipSuccess && e != hipErrorNotReady) {
                    scd_set_error("scd_kmeans_lloyd_run_sk: stream error while waiting for iteration %d: %s", i, hipGetErrorString(e));
                    return SCD_EHIP;
                }
                if (e == hipSuccess && host[7] != want) {
                    scd_set_error("scd_kmeans_lloyd_run_sk: iteration %d finished without publishing its statistics", i);
                    return SCD_EHIP;
                }
            }
        }
        refined_seen = host[3];
        changed_prev = changed_seen;
        changed_seen = host[4];
        if (host[5] > 0.) return 2;
        if (i > 0 && host[4] == 0.) return 1;          // strict: np.array_equal(labels, labels_old)
        return host[2] <= tol ? 1 : 0;                  // center_shift_tot <= tol
    };
    for (int it = 0;; ++it) {
        const float* c_in = it == 0 ? C_start : C_ring + (size_t)((it - 1) % 3) * kd;
        float* c_out = C_ring + (size_t)(it % 3) * kd;
        const bool few = it >= 2 && refined_seen >= 0. && refined_seen <= 64.;
        double pred = changed_seen;
        if (it >= 3 && changed_prev > 0. && changed_seen < changed_prev) pred = changed_seen * (changed_seen / changed_prev) * (changed_seen / changed_prev);
        const bool full = it < 2 || changed_seen < 0. || pred > many;
        const int flags = (few ? SCD_ESTEP_FEW : 0) | (it > 0 ? SCD_ESTEP_CENTRES_FROM_FINALIZE : 0) | (full ? SCD_LLOYD_FULL : 0);
        h->run_seq += 1.0;
        seq_of[it & 1] = h->run_seq;
        const int rc = lloyd_step_delta_impl(h, X, prep, n, X16, n, d, k, lab_ring + (size_t)(it % 3) * n, labels_prev, c_in, c_out, sums, counts,
                                             nullptr, nullptr, sumsq4, stats_ring + (it & 1) * 5, flags, ws_e, ws_e_bytes, ws_m, ws_m_bytes, stream,
                                             h->run_dev + (it & 1) * 8, h->run_seq, 1);
        if (rc) return rc;
        ++launched;
        delta_steps += full ? 0 : 1;
        if (it > 0) {
            const int s = settle(it - 1);
            if (s < 0) return s;
            if (s == 2) { status = 1; stop_it = it - 1; break; }
            if (s == 1 || it == max_iter) { stop_it = it - 1; break; }
        }
    }
    if (!status) {
        SCD_HIP(hipMemcpyAsync(final_labels, lab_ring + (size_t)((stop_it + 1) % 3) * n, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        SCD_HIP(hipMemcpyAsync(final_C, C_ring + (size_t)(stop_it % 3) * kd, kd * 4, hipMemcpyDeviceToDevice, st));
    }
    result_host[0] = (double)status;
    result_host[1] = (double)(stop_it + 1);
    result_host[2] = (double)delta_steps;
    result_host[3] = (double)launched;
    return SCD_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// scd_kpp_seed_lockstep_sharded: the rounds of KMeansEngine.kpp_lockstep over a ROW SHARD behind one call (one process per GPU, SURVEY
// 8e; the reference is single-process: sskm_constrained.py:28-44).  A round needs three exchanges - the shards' sums (-> the float32
// total that `prob = d2 / d2.sum()` divides by), the shards' probability masses (-> this shard's prefix), the candidate rows (-> the
// first owner's row) - each an all-gather of a few bytes per restart that the caller supplies as a callback, as the Lloyd loop's
// all-reduce (scd_kmeans_lloyd_run_sharded).  Same arithmetic, in the same order, as the Python-driven rounds it replaces
// (scd_amd/kmeans.py: kpp_lockstep), which remain as the A/B and for callers without a callback.
__global__ void __launch_bounds__(64) kss_total_kernel(const double* __restrict__ recv, int world, int R, double* __restrict__ tot) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    double s = 0.0;
    for (int w = 0; w < world; ++w) s += recv[(size_t)w * R + r];          // rank order: the same bits on every rank
    tot[r] = s;
}
__global__ void __launch_bounds__(64) kss_prefix_kernel(const double* __restrict__ recv, int rank, int R, double* __restrict__ pre) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= R) return;
    double s = 0.0;
    for (int w = 0; w < rank; ++w) s += recv[(size_t)w * R + r];
    pre[r] = s;
}
// block r: send[r] = [hit flag | the drawn row] (float32, 1 + d values; a shard without the hit sends row 0 with flag 0, as the mirror does)
__global__ void __launch_bounds__(256) kss_pack_kernel(const float* __restrict__ X, const long long* __restrict__ idx, int d,
                                                       float* __restrict__ send) {
    const int r = blockIdx.x;
    const long long i = idx[r];
    const float* row = X + (size_t)(i < 0 ? 0 : i) * d;
    float* out = send + (size_t)r * (1 + d);
    if (threadIdx.x == 0) out[0] = i >= 0 ? 1.f : 0.f;
    for (int j = threadIdx.x; j < d; j += 256) out[1 + j] = row[j];
}
// block r: the first rank (in rank order) that reports a hit owns restart r's new centre; none: rank 0's row and picks = -1
__global__ void __launch_bounds__(256) kss_select_kernel(const float* __restrict__ recv, int world, int R, int d, float* __restrict__ rows,
                                                         float* __restrict__ slot, long long ldc, long long* __restrict__ pick) {
    const int r = blockIdx.x;
    int owner = -1;
    for (int w = 0; w < world; ++w)
        if (recv[((size_t)w * R + r) * (1 + d)] > 0.f) { owner = w; break; }
    const float* src = recv + ((size_t)(owner < 0 ? 0 : owner) * R + r) * (1 + d) + 1;
    for (int j = threadIdx.x; j < d; j += 256) {
        const float v = src[j];
        rows[(size_t)r * d + j] = v;
        slot[(size_t)r * ldc + j] = v;
    }
    if (threadIdx.x == 0) pick[r] = owner < 0 ? -1 : 0;
}
static size_t kss_small_bytes(int d, int R) { return scd_align(8 * (size_t)R) * 3 + scd_align(8 * (size_t)R) + scd_align(4 * (size_t)R * d); }
extern "C" size_t scd_kpp_seed_sharded_ws_bytes(int64_t n, int d, int R) {
    return (size_t)R * scd_kpp_draw_ws_bytes(n) + scd_kpp_update_ws_bytes(n, d) + kss_small_bytes(d, R) + 256;
}
extern "C" size_t scd_kpp_seed_sharded_xbuf_bytes(int d, int R, int world) {
    const size_t unit = scd_align(4 * (size_t)R * (1 + d) > 8 * (size_t)R ? 4 * (size_t)R * (1 + d) : 8 * (size_t)R);
    return unit * (size_t)(1 + world);
}
extern "C" int scd_kpp_seed_lockstep_sharded(scd_handle h, const float* X, const void* X16, int64_t n, int d, int R, float* d2, int64_t ld,
                                             const float* r_dev, int T, float* C_buf, int k, int m0, int64_t* picks_out, void* ws,
                                             size_t ws_bytes, void* stream_, void* xbuf, size_t xbuf_bytes, scd_gather_fn gather,
                                             void* gather_ctx, int rank, int world) {
    SCD_DEVICE_ENTRY(h, "scd_kpp_seed_lockstep_sharded");
    SCD_REQUIRE(X && d2 && r_dev && C_buf && picks_out && ws && xbuf && gather && n > 0 && d > 0 && R > 0 && ld >= n && T >= 0 && m0 >= 1 &&
                    m0 + T <= k && world >= 1 && rank >= 0 && rank < world,
                "scd_kpp_seed_lockstep_sharded: bad arguments");
    SCD_REQUIRE(ws_bytes >= scd_kpp_seed_sharded_ws_bytes(n, d, R) && xbuf_bytes >= scd_kpp_seed_sharded_xbuf_bytes(d, R, world),
                "scd_kpp_seed_lockstep_sharded: workspace / exchange buffer too small");
    hipStream_t st = (hipStream_t)stream_;
    const size_t draw_nb = (size_t)R * scd_kpp_draw_ws_bytes(n), upd_nb = scd_kpp_update_ws_bytes(n, d);
    char* w = (char*)ws;
    void* draw_ws = w;
    void* upd_ws = w + draw_nb;
    double* tot = (double*)(w + draw_nb + upd_nb);
    double* pre = (double*)((char*)tot + scd_align(8 * (size_t)R));
    double* ps = (double*)((char*)pre + scd_align(8 * (size_t)R));
    long long* idx = (long long*)((char*)ps + scd_align(8 * (size_t)R));
    float* rows = (float*)((char*)idx + scd_align(8 * (size_t)R));
    const size_t unit = scd_kpp_seed_sharded_xbuf_bytes(d, R, world) / (size_t)(1 + world);
    char* send = (char*)xbuf;
    char* recv = send + unit;
    const long long ldc = (long long)k * d;
    static const int filt_env = getenv("SCD_KPP_FILTER") ? atoi(getenv("SCD_KPP_FILTER")) : 1;
    const bool filt = X16 && filt_env && muf_shape_ok(n, d, R);
    bool first_filter = true;
#define KSS_GATHER(BYTES)                                                                                        \
    {                                                                                                            \
        const int rc_ = gather(gather_ctx, send, recv, (int64_t)(BYTES), stream_);                               \
        if (rc_) {                                                                                               \
            scd_set_error("scd_kpp_seed_lockstep_sharded: the gather callback failed (%d)", rc_);                \
            return SCD_ERCCL;                                                                                    \
        }                                                                                                        \
    }
    for (int t = 0; t < T; ++t) {
        const float* r_t = r_dev + (size_t)t * R;
        // (1) shard sums -> total (float64 sums added in rank order; the draw rounds it to float32 as the reference's d2.sum())
        int rc = scd_sum_f32_multi(h, d2, n, ld, R, (double*)send, stream_);
        if (rc) return rc;
        KSS_GATHER(8 * (size_t)R)
        kss_total_kernel<<<(unsigned)scd_cdiv(R, 64), 64, 0, st>>>((const double*)recv, world, R, tot);
        // (2) shard probability masses -> this shard's prefix
        rc = scd_kpp_draw_multi(h, d2, n, ld, R, r_t, tot, nullptr, nullptr, (double*)send, draw_ws, draw_nb, stream_);
        if (rc) return rc;
        KSS_GATHER(8 * (size_t)R)
        kss_prefix_kernel<<<(unsigned)scd_cdiv(R, 64), 64, 0, st>>>((const double*)recv, rank, R, pre);
        // (3) the draw on this shard, its candidate rows -> the first owner's rows
        rc = scd_kpp_draw_multi(h, d2, n, ld, R, r_t, tot, pre, (int64_t*)idx, nullptr, draw_ws, draw_nb, stream_);
        if (rc) return rc;
        kss_pack_kernel<<<R, 256, 0, st>>>(X, idx, d, (float*)send);
        SCD_LAUNCH_CHECK();
        KSS_GATHER(4 * (size_t)R * (1 + d))
        kss_select_kernel<<<R, 256, 0, st>>>((const float*)recv, world, R, d, rows, C_buf + (size_t)(m0 + t) * d, ldc,
                                             (long long*)(picks_out + (size_t)t * R));
        SCD_LAUNCH_CHECK();
        if (t + 1 == T) break;
        // (4) d2 = min(d2, ||x - new centre||^2): through the MFMA filter once a new centre wins few rows, as the single-process loop
        if (filt && m0 + t >= 8) {
            rc = scd_kpp_update_filter(h, X16, n, d, R, rows, d2, ld, first_filter ? 1 : 0, upd_ws, upd_nb, stream_);
            first_filter = false;
        } else {
            rc = scd_kmeans_min_update_multi(h, X, rows, n, d, R, d2, ld, stream_);
        }
        if (rc) return rc;
    }
#undef KSS_GATHER
    return SCD_OK;
}

