// Transformer encoder towers for gfx950: CLIP ViT-B/16 visual and text (third-party `clip`, call sites
// /root/reference/main_unsup.py:127 and local_utils/clip_lang_util.py:101-102; structure in SURVEY.md appendix B)
// and the DINO/GCD ViT-B/16 of /root/reference/gcd/models/vision_transformer.py:135-219.
//
// Data layout: activations are token-major [image*T + token][width] fp16 (the residual stream is fp16 like the
// reference's `clip.load` model on GPU); LayerNorm statistics, softmax and every GEMM accumulate in fp32.
// Internally the batch is padded to a multiple of 128 images so every GEMM is an exact multiple of its
// 128x128x64 tile (197*128 rows).
#include "common.h"
#include "gemm.h"
#include <vector>
#include <stdlib.h>

struct scd_encoder {
    scd_encoder_desc d;
    std::vector<const void*> w;
    // optional HIP-event timing of the dominant kernel (the fc1 GEMM of every block), see scd_encoder_timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double timed_flop = 0.0;
    // LayerNorm folded into the QKV / fc1 GEMMs (run_blocks): per layer W' = W * gamma (fp16), colsum[n] = sum_k W'[n][k],
    // b'[n] = b[n] + sum_k beta[k] W[n][k]; one device allocation, owned here
    struct Folded { const half_t *wq, *w1; const float *csq, *bq, *cs1, *b1; };
    std::vector<Folded> folded;
    void* folded_mem = nullptr;
};

enum { W_PATCH = 0, W_PATCH_B = 1, W_CLS = 2, W_POS = 3, W_LNPRE_W = 4, W_LNPRE_B = 5, W_LNPOST_W = 6, W_LNPOST_B = 7,
       W_PROJ = 8, W_LAYER0 = 9, W_PER_LAYER = 12 };
enum { L_LN1_W = 0, L_LN1_B, L_QKV_W, L_QKV_B, L_PROJ_W, L_PROJ_B, L_LN2_W, L_LN2_B, L_FC1_W, L_FC1_B, L_FC2_W, L_FC2_B };

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; width % 256 == 0; lane owns 4-element groups (i*64+lane)*4
// st1 / st2 (may be NULL): this lane's share of the STORED row's {sum, sum of squares} (the fp16 values as written, fp32 accumulation,
// groups and elements ascending - the order of round 3's separate row_stats_kernel pass over x, whose bits these are)
__device__ __forceinline__ void stats_acc(const half4& o, float* st1, float* st2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float f = (float)o[j];
        *st1 += f;
        *st2 = fmaf(f, f, *st2);
    }
}
__device__ __forceinline__ void stats_store(float s1, float s2, long long* stats, long long r, int lane) {
    s1 = wave_sum_f32(s1);
    s2 = wave_sum_f32(s2);
    if (lane == 0) {   // the fixed-point format of scd_gemm_ln (gemm.h)
        stats[2 * r] = __float2ll_rn(s1 * 16777216.f);
        stats[2 * r + 1] = __float2ll_rn(s2 * 1048576.f);
    }
}
template <int MAXG>
__device__ __forceinline__ void ln_row(const float (&v)[MAXG][4], int groups, int width, float eps, const float* g,
                                       const float* b, half_t* out, int lane, float* st1 = nullptr, float* st2 = nullptr) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
        if (i < groups) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mean = wave_sum_f32(s) / (float)width;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
        if (i < groups) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q = fmaf(d, d, q);
            }
        }
    const float rstd = rsqrtf(wave_sum_f32(q) / (float)width + eps);
#pragma unroll
    for (int i = 0; i < MAXG; ++i)
        if (i < groups) {
            const int c = (i * 64 + lane) * 4;
            const float4 gg = *(const float4*)(g + c);
            const float4 bb = *(const float4*)(b + c);
            half4 o;
            o[0] = (half_t)((v[i][0] - mean) * rstd * gg.x + bb.x);
            o[1] = (half_t)((v[i][1] - mean) * rstd * gg.y + bb.y);
            o[2] = (half_t)((v[i][2] - mean) * rstd * gg.z + bb.z);
            o[3] = (half_t)((v[i][3] - mean) * rstd * gg.w + bb.w);
            *(half4*)(out + c) = o;
            if (st1) stats_acc(o, st1, st2);
        }
}

// rows: if row_index != NULL, input row = row_index[r] (gather of CLS / EOT rows)
__global__ void __launch_bounds__(256) layernorm_kernel(const half_t* __restrict__ x, const int* __restrict__ row_index,
                                                        long long rows, int width, float eps, const float* __restrict__ g,
                                                        const float* __restrict__ b, half_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long src = row_index ? row_index[r] : r;
    const int groups = width >> 8;
    float v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < groups) {
            const half4 h4 = *(const half4*)(x + src * width + (i * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = (float)h4[j];
        }
    ln_row<4>(v, groups, width, eps, g, b, out + r * width, lane);
}

// ------------------------------------------------------------------------------------------------ embeddings
// im2col for the stride-16 patch convolution: out[(b*np + p)][c*P*P + i*P + j] = img[b][c][py*P+i][px*P+j]
template <typename T>
__global__ void __launch_bounds__(256) im2col_kernel(const T* __restrict__ img, int batch, long long rows_pad, int image, int patch,
                                                     half_t* __restrict__ out) {
    const int gp = image / patch, np = gp * gp, kk = 3 * patch * patch;
    const long long total8 = rows_pad * kk / 8;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total8) return;
    const long long e = t * 8;
    const int col = (int)(e % kk);
    const long long row = e / kk;
    const int b = (int)(row / np), p = (int)(row % np);
    half8 o;
    if (b < batch) {
        const int c = col / (patch * patch), rem = col % (patch * patch), i = rem / patch, j = rem % patch;
        const int py = p / gp, px = p % gp;
        const T* src = img + (((size_t)b * 3 + c) * image + (py * patch + i)) * image + px * patch + j;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (half_t)(float)src[q];
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (half_t)0.f;
    }
    *(half8*)(out + e) = o;
}

// token assembly (+ optional ln_pre): x[b][0] = cls + pos[0]; x[b][1+p] = patch[b*np+p] + patch_bias + pos[1+p]
__global__ void __launch_bounds__(256) assemble_visual_kernel(const half_t* __restrict__ patch, const float* __restrict__ patch_b,
                                                              const float* __restrict__ cls, const float* __restrict__ pos,
                                                              long long rows, int batch, int T, int width, const float* __restrict__ lg,
                                                              const float* __restrict__ lb, float eps, half_t* __restrict__ out,
                                                              long long* __restrict__ stats) {
    // stats (may be NULL): the first block's LayerNorm statistics of the rows written here (round 3 ran a
    // kernel of its own over x for them)
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long b = r / T;
    const int t = (int)(r % T);
    const int groups = width >> 8;
    if (b >= batch) {   // padding rows (the row count is rounded up to the GEMM tile): zeros
        for (int c = lane * 4; c < width; c += 256) *(half4*)(out + r * width + c) = half4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if (stats && lane < 2) stats[2 * r + lane] = 0;
        return;
    }
    float s1 = 0.f, s2 = 0.f;
    float v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < groups) {
            const int c = (i * 64 + lane) * 4;
            const float4 pp = *(const float4*)(pos + (size_t)t * width + c);
            float4 a;
            if (t == 0) {
                a = *(const float4*)(cls + c);
            } else {
                const half4 h4 = *(const half4*)(patch + ((size_t)b * (T - 1) + (t - 1)) * width + c);
                a = make_float4((float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]);
                if (patch_b) {
                    const float4 pb = *(const float4*)(patch_b + c);
                    a.x += pb.x; a.y += pb.y; a.z += pb.z; a.w += pb.w;
                }
            }
            v[i][0] = a.x + pp.x; v[i][1] = a.y + pp.y; v[i][2] = a.z + pp.z; v[i][3] = a.w + pp.w;
        }
    if (lg) {
        ln_row<4>(v, groups, width, eps, lg, lb, out + r * width, lane, stats ? &s1 : nullptr, &s2);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < groups) {
                half4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (half_t)v[i][j];
                *(half4*)(out + r * width + (i * 64 + lane) * 4) = o;
                if (stats) stats_acc(o, &s1, &s2);
            }
    }
    if (stats) stats_store(s1, s2, stats, r, lane);
}

// The same rows, a wave taking ONE token position t of RB consecutive images: the position's embedding, the class token / patch bias and
// ln_pre's gamma / beta are loaded once per wave instead of once per row.  assemble_visual_kernel (one row per wave) read 10.5 KB per
// 1.5-KB row out of L2 - 856 us per 3,990-image launch, 2.8 TB/s of HBM traffic and ~10 TB/s of L2 traffic: the L2 was the bound.  Same
// arithmetic in the same order (ln_row's), so the rows and their statistics keep their bits.  Items: T x ceil(batch / RB) wave items in
// (image chunk, t) order - the four waves of a block write neighbouring rows -, then one item per zero padding row.
template <int RB>
__global__ void __launch_bounds__(256) assemble_visual_rows_kernel(const half_t* __restrict__ patch, const float* __restrict__ patch_b,
                                                                   const float* __restrict__ cls, const float* __restrict__ pos,
                                                                   long long rows, int batch, int T, int width, const float* __restrict__ lg,
                                                                   const float* __restrict__ lb, float eps, half_t* __restrict__ out,
                                                                   long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nbc = (batch + RB - 1) / RB;
    const long long n_main = (long long)T * nbc;
    const int groups = width >> 8;
    if (item >= n_main) {   // padding rows (the row count is rounded up to the GEMM tile): zeros
        const long long r = (long long)batch * T + (item - n_main);
        if (r >= rows) return;
        for (int c = lane * 4; c < width; c += 256) *(half4*)(out + r * width + c) = half4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if (stats && lane < 2) stats[2 * r + lane] = 0;
        return;
    }
    const int t = (int)(item % T);
    const int b0 = (int)(item / T) * RB;
    float4 pp[4], ca[4], gg[4], bb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < groups) {
            const int c = (i * 64 + lane) * 4;
            pp[i] = *(const float4*)(pos + (size_t)t * width + c);
            ca[i] = t == 0 ? *(const float4*)(cls + c) : (patch_b ? *(const float4*)(patch_b + c) : make_float4(0.f, 0.f, 0.f, 0.f));
            if (lg) { gg[i] = *(const float4*)(lg + c); bb[i] = *(const float4*)(lb + c); }
        }
    // all RB rows' patch values first: 3 x RB loads in flight per wave before the first reduction (one row at a time, a wave had 1.5 KB
    // in flight and the launch ran at the latency of its dependent load -> reduce -> store chains)
    half4 hv[RB][4];
    if (t != 0) {
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            const int b = b0 + q < batch ? b0 + q : batch - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < groups) hv[q][i] = *(const half4*)(patch + ((size_t)b * (T - 1) + (t - 1)) * width + (i * 64 + lane) * 4);
        }
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
        const int b = b0 + q;
        if (b >= batch) break;
        const long long r = (long long)b * T + t;
        float v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < groups) {
                float4 a;
                if (t == 0) {
                    a = ca[i];
                } else {
                    const half4 h4 = hv[q][i];
                    a = make_float4((float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]);
                    if (patch_b) { a.x += ca[i].x; a.y += ca[i].y; a.z += ca[i].z; a.w += ca[i].w; }
                }
                v[i][0] = a.x + pp[i].x; v[i][1] = a.y + pp[i].y; v[i][2] = a.z + pp[i].z; v[i][3] = a.w + pp[i].w;
            }
        float s1 = 0.f, s2 = 0.f;
        half_t* o_row = out + r * width;
        if (lg) {   // ln_row's arithmetic with gamma / beta already in registers
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < groups) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            const float mean = wave_sum_f32(s) / (float)width;
            float qq = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < groups) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dd = v[i][j] - mean;
                        qq = fmaf(dd, dd, qq);
                    }
                }
            const float rstd = rsqrtf(wave_sum_f32(qq) / (float)width + eps);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < groups) {
                    half4 o;
                    o[0] = (half_t)((v[i][0] - mean) * rstd * gg[i].x + bb[i].x);
                    o[1] = (half_t)((v[i][1] - mean) * rstd * gg[i].y + bb[i].y);
                    o[2] = (half_t)((v[i][2] - mean) * rstd * gg[i].z + bb[i].z);
                    o[3] = (half_t)((v[i][3] - mean) * rstd * gg[i].w + bb[i].w);
                    *(half4*)(o_row + (i * 64 + lane) * 4) = o;
                    if (stats) stats_acc(o, &s1, &s2);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < groups) {
                    half4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = (half_t)v[i][j];
                    *(half4*)(o_row + (i * 64 + lane) * 4) = o;
                    if (stats) stats_acc(o, &s1, &s2);
                }
        }
        if (stats) stats_store(s1, s2, stats, r, lane);
    }
}

// text: x[b][t] = tok_emb[token] + pos[t]; eot_row[b] = b*T + argmax_t token (first maximum, like torch.argmax)
// S = positions per row of `tokens` (77), T <= S = positions computed; the EOT position (argmax over all S) must be < T
__global__ void __launch_bounds__(256) embed_text_kernel(const int* __restrict__ tokens, int batch, const half_t* __restrict__ emb,
                                                         int vocab, const float* __restrict__ pos, long long rows, int T, int S,
                                                         int width, half_t* __restrict__ out, int* __restrict__ eot_row,
                                                         long long* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const long long b = r / T;
    const int t = (int)(r % T);
    int tok = 0;
    if (b < batch) tok = tokens[b * S + t];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < width; c += 256) {
        const half4 e4 = *(const half4*)(emb + (size_t)tok * width + c);
        const float4 pp = *(const float4*)(pos + (size_t)t * width + c);
        half4 o;
        o[0] = (half_t)((float)e4[0] + pp.x); o[1] = (half_t)((float)e4[1] + pp.y);
        o[2] = (half_t)((float)e4[2] + pp.z); o[3] = (half_t)((float)e4[3] + pp.w);
        *(half4*)(out + r * width + c) = o;
        if (stats) stats_acc(o, &s1, &s2);
    }
    if (stats) stats_store(s1, s2, stats, r, lane);          // the first block's LayerNorm statistics
    if (t == 0 && lane == 0) {
        int best = 0, bt = -2147483647;
        if (b < batch) {
            for (int i = 0; i < S; ++i) {
                const int v = tokens[b * S + i];
                if (v > bt) { bt = v; best = i; }
            }
            eot_row[b] = (int)(b * T + (best < T ? best : T - 1));
        }
    }
}

// rows[i] = i*T for the first `batch` entries (when first_only is 0), row 0 for the padding entries batch..n-1
__global__ void cls_rows_kernel(int* rows, int n, int T, int batch, int pad_only) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i >= batch) rows[i] = 0;
    else if (!pad_only) rows[i] = i * T;
}

// copy the first `batch` rows to the caller, optionally L2-normalised (F.normalize semantics)
__global__ void __launch_bounds__(256) emit_kernel(const half_t* __restrict__ in, int batch, int dim, int normalize,
                                                   half_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= batch) return;
    float s = 0.f;
    for (int j = lane; j < dim; j += 64) {
        const float v = (float)in[(size_t)r * dim + j];
        s = fmaf(v, v, s);
    }
    s = wave_sum_f32(s);
    const float inv = normalize ? 1.0f / fmaxf(sqrtf(s), 1e-12f) : 1.0f;
    for (int j = lane; j < dim; j += 64) out[(size_t)r * dim + j] = (half_t)((float)in[(size_t)r * dim + j] * inv);
}

// ------------------------------------------------------------------------------------------------ attention
// One block (4 waves) per (image, head); head_dim = 64.  K (XOR-swizzled 128-B rows) and V (row-major, 192-B rows)
// live in LDS; a wave owns 32-query blocks.  S^T = K Q^T with v_mfma_f32_32x32x16_f16 (A = K rows, B = Q rows):
// lane = query, registers = keys, so the softmax reductions are in-lane plus one exchange between lanes l and l+32;
// the exponentiated S^T registers are then directly the B operand of O^T = V^T P^T (k order of the accumulator
// layout: key = 16s + 8(j>>2) + 4h + (j&3)), and the matching V^T A-fragments come from the hardware transposing
// read ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group), so V is staged with plain 16-byte row copies.
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));

template <int NB>   // NB = ceil(T/32): 7 for T=197, 3 for T=77
__global__ void __launch_bounds__(256, 2) attention_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int T, int width,
                                                           int heads, int causal, int xmode) {
    constexpr int TP = NB * 32;
    constexpr int VS = 192;                          // V row stride in bytes: 4 rows x 64 B of a tr-read tile the 64 banks
    __shared__ __attribute__((aligned(16))) char kl[TP * 128];
    __shared__ __attribute__((aligned(16))) char vl[TP * VS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, hh = lane >> 5;
    const int img = blockIdx.x / heads, head = blockIdx.x % heads;
    const size_t row0 = (size_t)img * T;
    const int ld = 3 * width;
    const half_t* qbase = qkv + row0 * ld + head * 64;
    const half_t* kbase = qbase + width;
    const half_t* vbase = qbase + 2 * width;

    // stage K and V: all 2*NB 16-byte loads of a thread are issued before the first LDS write (one memory round trip)
    {
        uint4 kv[NB], vv[NB];
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int i = it * 256 + tid, row = i >> 3, ch = i & 7;
            kv[it] = make_uint4(0, 0, 0, 0);
            vv[it] = make_uint4(0, 0, 0, 0);
            if (row < T) {
                kv[it] = *(const uint4*)(kbase + (size_t)row * ld + 8 * ch);
                vv[it] = *(const uint4*)(vbase + (size_t)row * ld + 8 * ch);
            }
        }
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int i = it * 256 + tid, row = i >> 3, ch = i & 7;
            *(uint4*)(kl + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = kv[it];
            *(uint4*)(vl + row * VS + ch * 16) = vv[it];
        }
    }
    __syncthreads();

    // transposing-read address of this lane inside a (4 keys x 16 d) tile: group g = lane>>4 selects d half and key half
    const int L = lane & 15;
    const int tr_off = (4 * hh + (L >> 2)) * VS + (16 * ((lane >> 4) & 1) + 4 * (L & 3)) * 2;

    for (int qb = wave; qb < NB; qb += 4) {
        if (xmode & 4) break;
        const int query = qb * 32 + r;
        const int qrow = query < T ? query : T - 1;
        half8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const half8*)(qbase + (size_t)qrow * ld + 16 * s + 8 * hh);
        f32x16 sacc[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[kb][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = kb * 32 + r;
                const half8 kf = *(const half8*)(kl + row * 128 + (((2 * s + hh) ^ ((row >> 1) & 7)) << 4));
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kb], 0, 0, 0);
            }
        }
        // softmax over keys on the raw scores: only the last key block can hold padded keys; the 1/sqrt(64) scale is
        // folded into the exp2 argument: p = 2^((s - max) * 0.125 * log2 e)
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = sacc[kb][i];
                if (kb == NB - 1 || causal) {
                    const int key = kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    if (key >= T || (causal && key > query)) v = -INFINITY;
                    sacc[kb][i] = v;
                }
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float cs = 0.125f * 1.4426950408889634f;
        const float mxs = mx * cs;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kb][i], cs, -mxs));
                sacc[kb][i] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x16 oacc[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            if (xmode & 2) break;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                half8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[kb][8 * s + j];
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* base = vl + (kb * 32 + 16 * s) * VS + db * 64 + tr_off;
                    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base));
                    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 8 * VS));
                    const half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
                    half8 vf;
                    vf[0] = l4[0]; vf[1] = l4[1]; vf[2] = l4[2]; vf[3] = l4[3];
                    vf[4] = h4[0]; vf[5] = h4[1]; vf[6] = h4[2]; vf[7] = h4[3];
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
        }
        if (query < T) {
            half_t* orow = out + (row0 + query) * width + head * 64;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 o;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) o[q4] = (half_t)(oacc[db][4 * g + q4] * inv);
                    *(half4*)(orow + db * 32 + 8 * g + 4 * hh) = o;
                }
        }
    }
}

// Short contexts (T <= 32: the text tower after context trimming, 10-20 tokens): ONE WAVE per (prompt, head), four items per block.
// attention_kernel<1> gave such an item a whole 256-thread block of which three waves only helped to stage 2 x 2 KB and then idled: at
// 20,480 prompts x 8 heads a launch was 163,840 blocks of almost no work, 383 us for 1.3 GB (round 6: 16 % of the vocabulary build).
// Here a wave stages its item's K and V into its own LDS region (no block barrier: a wave's LDS operations execute in order) and runs the
// one 32-query block exactly as attention_kernel does - the same MFMAs on the same operands, so the same bits.
__global__ void __launch_bounds__(256) attention_short_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int T, int width,
                                                              int heads, int causal, int items) {
    constexpr int VS = 192;
    __shared__ __attribute__((aligned(16))) char kl_all[4][32 * 128];
    __shared__ __attribute__((aligned(16))) char vl_all[4][32 * VS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= items) return;
    char* kl = kl_all[wave];
    char* vl = vl_all[wave];
    const int r = lane & 31, hh = lane >> 5;
    const int img = item / heads, head = item % heads;
    const size_t row0 = (size_t)img * T;
    const int ld = 3 * width;
    const half_t* qbase = qkv + row0 * ld + head * 64;
    const half_t* kbase = qbase + width;
    const half_t* vbase = qbase + 2 * width;
    {   // 32 rows x 8 chunks of 16 bytes each for K and V: four per lane, all loads issued before the first LDS write
        uint4 kv[4], vv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = it * 64 + lane, row = i >> 3, ch = i & 7;
            kv[it] = make_uint4(0, 0, 0, 0);
            vv[it] = make_uint4(0, 0, 0, 0);
            if (row < T) {
                kv[it] = *(const uint4*)(kbase + (size_t)row * ld + 8 * ch);
                vv[it] = *(const uint4*)(vbase + (size_t)row * ld + 8 * ch);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = it * 64 + lane, row = i >> 3, ch = i & 7;
            *(uint4*)(kl + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4)) = kv[it];
            *(uint4*)(vl + row * VS + ch * 16) = vv[it];
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): this wave's LDS writes have landed (no other wave touches the region)
    __builtin_amdgcn_wave_barrier();
    const int L = lane & 15;
    const int tr_off = (4 * hh + (L >> 2)) * VS + (16 * ((lane >> 4) & 1) + 4 * (L & 3)) * 2;
    const int query = r;
    const int qrow = query < T ? query : T - 1;
    half8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const half8*)(qbase + (size_t)qrow * ld + 16 * s + 8 * hh);
    f32x16 sacc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int row = r;
        const half8 kf = *(const half8*)(kl + row * 128 + (((2 * s + hh) ^ ((row >> 1) & 7)) << 4));
        sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc, 0, 0, 0);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = sacc[i];
        const int key = (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (key >= T || (causal && key > query)) v = -INFINITY;
        sacc[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float cs = 0.125f * 1.4426950408889634f;
    const float mxs = mx * cs;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(fmaf(sacc[i], cs, -mxs));
        sacc[i] = p;
        sum += p;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = __builtin_amdgcn_rcpf(sum);
    f32x16 oacc[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[db][i] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        half8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[8 * s + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const char* base = vl + (16 * s) * VS + db * 64 + tr_off;
            const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base));
            const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 8 * VS));
            const half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
            half8 vf;
            vf[0] = l4[0]; vf[1] = l4[1]; vf[2] = l4[2]; vf[3] = l4[3];
            vf[4] = h4[0]; vf[5] = h4[1]; vf[6] = h4[2]; vf[7] = h4[3];
            oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[db], 0, 0, 0);
        }
    }
    if (query < T) {
        half_t* orow = out + (row0 + query) * width + head * 64;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 o;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) o[q4] = (half_t)(oacc[db][4 * g + q4] * inv);
                *(half4*)(orow + db * 32 + 8 * g + 4 * hh) = o;
            }
    }
}

// Persistent attention for the ViT towers (T <= 224 tokens, head_dim 64): one 8-wave block per CU loops over its
// (image, head) items.  Wave 7 is the producer: it streams the NEXT item's K and V head slices into the other half of a
// double-buffered LDS image with LDS-DMA (8 rows x 128 B per instruction, the bank swizzles applied to the per-lane source
// address) while waves 0-6 each compute one 32-query block of the current item exactly as attention_kernel does (S^T = K Q^T,
// in-lane softmax, O^T = V^T P^T through ds_read_b64_tr_b16); their Q fragments for the next item are prefetched into
// registers at the start of the current one.  One barrier per item.  ~85 KB per CU stay in flight, which is what the qkv
// stream (the kernel's floor: 620 MB per call) needs.
// LDS image per buffer: K 224 rows x 128 B, chunk c of row r at c ^ ((r>>1)&7); V 224 rows x 128 B, its 64-byte halves
// swapped when (r>>1)&1, which makes the transposing reads (4 rows x 64 B per 32-lane group) tile the 64 banks.  Rows past
// T repeat row T-1 (finite values); their scores are masked to -inf, so their P is exactly 0.
// T197 (T = 197, the ViT-B/16 towers): what the padding to 224 keys costs is left out at compile time - of the last key block only keys
// 192..196 can be valid, so 12 of its 16 exponentials per lane, its second pair of P V MFMAs (keys 208..223) with their V reads, and the
// ring fills of K rows >= 200 / V rows >= 208 are not executed (their probabilities are exactly 0 and 0 x finite adds +0 to a sum that
// starts at +0: the same bits).  The K rows that are not filled hold whatever the LDS held; their scores are never read.
template <bool T197>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
attention_persist_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out, int T, int width, int heads, int items, int xmode) {
    constexpr int NB = 7, TP = NB * 32, KV = TP * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K | V]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = 3 * width;
    const int n_my = (items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (n_my <= 0) return;
    if (wave == 7) {
        const unsigned sbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const int drow = lane >> 3, pc = lane & 7;
        auto issue_item = [&](int item, int buf) {
            const int img = item / heads, head = item - img * heads;
            const half_t* kb = qkv + (size_t)img * T * ld + width + head * 64;
            const half_t* vb = kb + width;
#pragma unroll 4
            for (int p = 0; p < (T197 ? 26 : TP / 8); ++p) {
                const int row = p * 8 + drow;
                const int rowc = row < T ? row : T - 1;
                const unsigned voff_k = (unsigned)(rowc * ld + ((pc ^ ((row >> 1) & 7)) << 3)) * 2;
                const unsigned voff_v = (unsigned)(rowc * ld + ((pc ^ (((row >> 1) & 1) << 2)) << 3)) * 2;
                const unsigned lds = sbase + buf * 2 * KV + p * 1024;
                if (!T197 || p < 25)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff_k), "s"(kb) : "memory");
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds + KV), "v"(voff_v), "s"(vb) : "memory");
            }
        };
        issue_item(blockIdx.x, 0);
        for (int i = 0; i < n_my; ++i) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // item i has landed
            __builtin_amdgcn_s_barrier();                               // consumers start item i; nobody reads item i-1's buffer any more
            asm volatile("" ::: "memory");
            if (i + 1 < n_my) issue_item(blockIdx.x + (i + 1) * gridDim.x, (i + 1) & 1);
        }
        return;
    }
    // ---- consumers: wave = query block
    const int r = lane & 31, hh = lane >> 5, L = lane & 15;
    const int query = wave * 32 + r;
    const int qrow = query < T ? query : T - 1;
    const int vswz = ((L >> 3) & 1) * 64;
    const int tr_off = (4 * hh + (L >> 2)) * 128 + 32 * ((lane >> 4) & 1) + 8 * (L & 3);
    half8 qf[4], qn[4];
    {
        const int img = blockIdx.x / heads, head = blockIdx.x - img * heads;
        const half_t* qbase = qkv + (size_t)img * T * ld + head * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const half8*)(qbase + (size_t)qrow * ld + 16 * s + 8 * hh);
    }
    for (int i = 0; i < n_my; ++i) {
        const int item = blockIdx.x + i * gridDim.x;
        const int img = item / heads, head = item - img * heads;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (xmode & 16) {          // timing ablation: no Q loads
#pragma unroll
            for (int s = 0; s < 4; ++s) qn[s] = qf[s];
        } else if (i + 1 < n_my) {
            const int item2 = item + gridDim.x;
            const int img2 = item2 / heads, head2 = item2 - img2 * heads;
            const half_t* qbase = qkv + (size_t)img2 * T * ld + head2 * 64;
#pragma unroll
            for (int s = 0; s < 4; ++s) qn[s] = *(const half8*)(qbase + (size_t)qrow * ld + 16 * s + 8 * hh);
        }
        const char* kl = smem + (i & 1) * 2 * KV;
        const char* vl = kl + KV;
        if (xmode & 4) {   // timing ablation: no compute at all
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[s] = qn[s];
            continue;
        }
        f32x16 sacc[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = kb * 32 + r;
                const half8 kf = *(const half8*)(kl + row * 128 + (((2 * s + hh) ^ ((row >> 1) & 7)) << 4));
                sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kb], 0, 0, 0);
            }
        }
        // softmax over keys on the raw scores (only the last key block holds padded keys); 1/sqrt(64) folded into exp2
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < (T197 ? 4 : 16); ++e) {   // only the last key block can hold padded keys (T197: its registers 4..15 always do)
            const int key = (NB - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
            if (key >= T) sacc[NB - 1][e] = -INFINITY;
        }
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int e = 0; e < (T197 && kb == NB - 1 ? 4 : 16); e += 2) mx = fmaxf(fmaxf(mx, sacc[kb][e]), sacc[kb][e + 1]);   // v_max3_f32
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // p = 2^((s - max) * 0.125 * log2 e), two keys per instruction wherever the ISA has a packed form (v_pk_fma_f32,
        // v_pk_add_f32, v_cvt_pk_f16_f32): the kernel is VALU-bound (two consumer waves per SIMD, ~600 VALU instructions per
        // item and wave before this), not MFMA- or HBM-bound.  The probabilities are kept as packed fp16 - exactly the PV operand.
        typedef float f2v __attribute__((ext_vector_type(2)));
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const float cs = 0.125f * 1.4426950408889634f;
        const f2v cs2 = {cs, cs}, nmx2 = {-mx * cs, -mx * cs};
        f2v sum2 = {0.f, 0.f};
        h2v ph[NB][8];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (T197 && kb == NB - 1 && e >= 2) { ph[kb][e] = h2v{(_Float16)0.f, (_Float16)0.f}; continue; }   // keys >= 200: p = 0
                f2v t = {sacc[kb][2 * e], sacc[kb][2 * e + 1]};
                t = t * cs2 + nmx2;
                t.x = __builtin_amdgcn_exp2f(t.x);
                t.y = __builtin_amdgcn_exp2f(t.y);
                sum2 += t;
                ph[kb][e] = __builtin_convertvector(t, h2v);
            }
        float sum = sum2.x + sum2.y;
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x16 oacc[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
            for (int s = 0; s < (T197 && kb == NB - 1 ? 1 : 2); ++s) {   // T197: keys 208..223 contribute exactly nothing
                half8 pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pf[2 * j] = ph[kb][4 * s + j].x; pf[2 * j + 1] = ph[kb][4 * s + j].y; }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* base = vl + (kb * 32 + 16 * s) * 128 + ((db * 64) ^ vswz) + tr_off;
                    const fp16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base));
                    const fp16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(base + 8 * 128));
                    const half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
                    half8 vf;
                    vf[0] = l4[0]; vf[1] = l4[1]; vf[2] = l4[2]; vf[3] = l4[3];
                    vf[4] = h4[0]; vf[5] = h4[1]; vf[6] = h4[2]; vf[7] = h4[3];
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
        }
        // O leaves as whole 128-byte head slices: lane (query r, half hh) holds eight 8-byte pieces of its row, scattered over the
        // row's eight 16-byte chunks; written straight from there every store instruction touched 32 rows with 16 bytes each
        // (8 partial writes per cache line).  Through a per-wave LDS patch [32 rows][128 B] (chunk c of row r at c ^ (r & 7)) eight
        // consecutive lanes write one row's slice, four 1-KB instructions per wave and item.
        if (!(xmode & 8)) {
            char* op = smem + 2 * 2 * KV + wave * 4096;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4 o;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) o[q4] = (half_t)(oacc[db][4 * g + q4] * inv);
                    *(half4*)(op + r * 128 + (((db * 4 + g) ^ (r & 7)) << 4) + 8 * hh) = o;
                }
            asm volatile("" ::: "memory");      // the patch is re-read through another vector type: no compiler reordering across
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int row = (lane >> 3) + 8 * i4, c = lane & 7;
                const uint4 v4 = *(const uint4*)(op + row * 128 + ((c ^ (row & 7)) << 4));
                const int qy = wave * 32 + row;
                if (qy < T) *(uint4*)(out + ((size_t)img * T + qy) * width + head * 64 + c * 8) = v4;
            }
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = qn[s];
    }
}

// ------------------------------------------------------------------------------------------------ LayerNorm folding
// one wave per output row n: Wf[n][k] = fp16(W[n][k] * gamma[k]); colsum[n] = sum_k Wf[n][k]; biasf[n] = bias[n] + sum_k beta[k] W[n][k]
__global__ void __launch_bounds__(256) fold_ln_kernel(const half_t* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ bias, int N, int K,
                                                      half_t* __restrict__ Wf, float* __restrict__ colsum, float* __restrict__ biasf) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float cs = 0.f, bb = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = (float)W[(size_t)n * K + k];
        const half_t wf = (half_t)(w * gamma[k]);
        Wf[(size_t)n * K + k] = wf;
        cs += (float)wf;
        bb = fmaf(beta[k], w, bb);
    }
    cs = wave_sum_f32(cs);
    bb = wave_sum_f32(bb);
    if (lane == 0) {
        colsum[n] = cs;
        biasf[n] = bias[n] + bb;
    }
}

// dst[i] = x[rows[i]] and, when stats != null, dst_stats[i] = stats[rows[i]] (one wave per row)
__global__ void __launch_bounds__(256) gather_rows_stats_kernel(const half_t* __restrict__ x, const long long* __restrict__ stats,
                                                                const int* __restrict__ rows, int n, int width,
                                                                half_t* __restrict__ dst, long long* __restrict__ dst_stats) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const size_t src = (size_t)rows[i] * width;
    for (int c = lane * 8; c < width; c += 512) *(half8*)(dst + (size_t)i * width + c) = *(const half8*)(x + src + c);
    if (stats && lane < 2) dst_stats[2 * i + lane] = stats[2 * (size_t)rows[i] + lane];
}

// Attention of ONE query row per (image, head) - the last block only needs the CLS / EOT row (see run_blocks).  One wave per
// item: lanes = keys for the scores (each lane dots its keys' 128-byte K rows with the query), then lanes = head dims for
// O = P V with P broadcast from LDS.  Same scaling, fp16-rounded probabilities and fp32 accumulation as the block kernels.
// kv: [rows][2*width] (K | V), q: [n_img][width], out: [n_img][width]; qpos[img] = query position (causal limit) or null.
__global__ void __launch_bounds__(256) attention_single_query_kernel(const half_t* __restrict__ kv, const half_t* __restrict__ q,
                                                                     const int* __restrict__ qrow, half_t* __restrict__ out, int T,
                                                                     int width, int heads, int items, int causal) {
    __shared__ float ps[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= items) return;
    const int img = item / heads, head = item - img * heads;
    const int ldk = 2 * width;
    const half_t* kbase = kv + (size_t)img * T * ldk + head * 64;
    const half_t* vbase = kbase + width;
    const int limit = causal ? (qrow[img] - img * T) : T - 1;      // last key the query may see
    const int nkeys = (limit < T - 1 ? limit : T - 1) + 1;
    // lanes = (key sub-index ks, 16-byte chunk dc of the 128-byte head row): every load instruction covers 8 whole rows
    const int ks = lane >> 3, dc = lane & 7;
    const half8 q8 = *(const half8*)(q + (size_t)img * width + head * 64 + 8 * dc);
    for (int key0 = 0; key0 < nkeys; key0 += 8) {
        const int key = key0 + ks;
        const int kc = key < nkeys ? key : nkeys - 1;
        const half8 k8 = *(const half8*)(kbase + (size_t)kc * ldk + 8 * dc);
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf((float)k8[e], (float)q8[e], d);
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (dc == 0 && key < nkeys) ps[wv][key] = d;
    }
    __builtin_amdgcn_s_waitcnt(0);      // this wave's own LDS traffic only (one wave per item)
    float sc[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int key = lane + 64 * j;
        sc[j] = key < nkeys ? ps[wv][key] : -INFINITY;
        mx = fmaxf(mx, sc[j]);
    }
    mx = wave_max_f32(mx);
    const float cs = 0.125f * 1.4426950408889634f;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sc[j], cs, -mx * cs));     // exp2(-inf) = 0 for masked keys
        sum += pv;
        ps[wv][lane + 64 * j] = (float)(half_t)pv;
    }
    sum = wave_sum_f32(sum);
    const float inv = __builtin_amdgcn_rcpf(sum);
    __builtin_amdgcn_s_waitcnt(0);
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = 0.f;
    for (int key = ks; key < nkeys; key += 8) {
        const half8 v8 = *(const half8*)(vbase + (size_t)key * ldk + 8 * dc);
        const float pk = ps[wv][key];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = fmaf(pk, (float)v8[e], o8[e]);
    }
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = o8[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        o[e] = (half_t)(v * inv);
    }
    if (ks == 0) *(half8*)(out + (size_t)img * width + head * 64 + 8 * dc) = o;
}

// dst_a[i] = a[rows[i]], dst_b[i] = b[rows[i]] (one wave per row, width % 256 == 0)
__global__ void __launch_bounds__(256) gather2_rows_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                           const int* __restrict__ rows, int n, int width,
                                                           half_t* __restrict__ dst_a, half_t* __restrict__ dst_b) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const size_t src = (size_t)rows[i] * width, dst = (size_t)i * width;
    for (int c = lane * 8; c < width; c += 512) {
        *(half8*)(dst_a + dst + c) = *(const half8*)(a + src + c);
        *(half8*)(dst_b + dst + c) = *(const half8*)(b + src + c);
    }
}

// ------------------------------------------------------------------------------------------------ host side
// Row counts are padded to the GEMM tile (256), not the image count: with 197 tokens per image a whole-image padding would
// need multiples of 256 images.  rows: token rows; prows: patch rows (im2col); bh: rows of the CLS / EOT head.
struct EncPad {
    int batch, bh;
    int tokens;          // positions per sequence that are computed: d.tokens, or fewer for trimmed text (scd_clip_encode_text_len)
    long long rows, prows;
};
static inline EncPad make_pad(const scd_encoder_desc& d, int batch, int tokens = 0) {
    EncPad p;
    p.batch = batch;
    p.tokens = tokens > 0 ? tokens : d.tokens;
    p.bh = (batch + 255) / 256 * 256;
    p.rows = ((long long)batch * p.tokens + 255) / 256 * 256;
    p.prows = ((long long)batch * (p.tokens - 1) + 255) / 256 * 256;
    return p;
}

struct EncWs {
    half_t *x, *y, *qkv, *h, *cls, *outp;
    int* rows;
    long long *stats_a, *stats_b;   // [rows][2] fixed-point row sums of x for the folded LayerNorms (LN1 / LN2 input)
    half_t *xsel, *ysel, *hsel;     // the last block's CLS / EOT rows only: [bh][width], [bh][width], [bh][mlp_dim]
    long long* stats_sel;
    float* rs;                      // [rows][2] {rstd, -mean * rstd}: what the next LayerNorm-folded GEMM reads (scd_gemm_ln_finish)
    size_t total;
};
static EncWs carve(const scd_encoder_desc& d, const EncPad& pad, char* base) {
    EncWs w;
    const size_t rows = (size_t)pad.rows;
    const int bp = pad.bh;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += scd_align(bytes); return p; };
    w.x = (half_t*)take(rows * d.width * 2);
    w.y = (half_t*)take(rows * d.width * 2);
    w.qkv = (half_t*)take(rows * 3 * d.width * 2);
    w.h = (half_t*)take(rows * d.mlp_dim * 2);
    w.cls = (half_t*)take((size_t)bp * d.width * 2);
    w.outp = (half_t*)take((size_t)bp * (d.out_dim > 0 ? d.out_dim : d.width) * 2);
    w.rows = (int*)take((size_t)bp * 4);
    w.stats_a = (long long*)take(rows * 16);
    w.stats_b = (long long*)take(rows * 16);
    w.xsel = (half_t*)take((size_t)bp * d.width * 2);
    w.ysel = (half_t*)take((size_t)bp * d.width * 2);
    w.hsel = (half_t*)take((size_t)bp * d.mlp_dim * 2);
    w.stats_sel = (long long*)take((size_t)bp * 16);
    w.rs = (float*)take(rows * 8);
    w.total = off + 256;
    return w;
}

extern "C" int scd_encoder_create(scd_handle h, const scd_encoder_desc* desc, const void* const* weights, int n_weights,
                                  scd_encoder** out) {
    SCD_DEVICE_ENTRY(h, "scd_encoder_create");
    SCD_REQUIRE(h && desc && weights && out, "scd_encoder_create: null argument");
    const scd_encoder_desc& d = *desc;
    SCD_REQUIRE(d.kind >= 0 && d.kind <= 2, "scd_encoder_create: bad kind %d", d.kind);
    SCD_REQUIRE(d.width % 256 == 0 && d.width <= 1024 && d.width == d.heads * 64, "scd_encoder_create: width %d / heads %d (head_dim must be 64)", d.width, d.heads);
    SCD_REQUIRE(d.mlp_dim % 128 == 0 && d.layers > 0, "scd_encoder_create: bad mlp_dim/layers");
    SCD_REQUIRE(d.out_dim == 0 || d.out_dim % 128 == 0, "scd_encoder_create: out_dim %d must be a multiple of 128", d.out_dim);
    SCD_REQUIRE(n_weights == W_LAYER0 + W_PER_LAYER * d.layers, "scd_encoder_create: expected %d weight pointers, got %d",
                W_LAYER0 + W_PER_LAYER * d.layers, n_weights);
    if (d.kind == 1) {
        SCD_REQUIRE(d.tokens == 77 && d.vocab > 0, "scd_encoder_create: text tower expects 77 tokens");
        SCD_REQUIRE(weights[W_PATCH] && weights[W_POS] && weights[W_LNPOST_W] && weights[W_LNPOST_B] && weights[W_PROJ] && d.out_dim > 0,
                    "scd_encoder_create: text tower weight missing");
    } else {
        SCD_REQUIRE(d.patch > 0 && d.image % d.patch == 0 && (3 * d.patch * d.patch) % 64 == 0, "scd_encoder_create: bad patch/image");
        const int np = (d.image / d.patch) * (d.image / d.patch);
        SCD_REQUIRE(d.tokens == np + 1 && d.tokens == 197, "scd_encoder_create: visual tower expects 197 tokens, got %d", d.tokens);
        SCD_REQUIRE(weights[W_PATCH] && weights[W_CLS] && weights[W_POS] && weights[W_LNPOST_W] && weights[W_LNPOST_B],
                    "scd_encoder_create: visual tower weight missing");
        SCD_REQUIRE((d.out_dim > 0) == (weights[W_PROJ] != nullptr), "scd_encoder_create: projection / out_dim mismatch");
    }
    for (int l = 0; l < d.layers; ++l)
        for (int j = 0; j < W_PER_LAYER; ++j)
            SCD_REQUIRE(weights[W_LAYER0 + l * W_PER_LAYER + j], "scd_encoder_create: layer %d weight %d is null", l, j);
    scd_encoder* e = new scd_encoder();
    e->d = d;
    e->w.assign(weights, weights + n_weights);
    // fold LN1 into the QKV weights and LN2 into the fc1 weights (used when every GEMM of a block can take the four-wave
    // kernel: widths multiples of 256)
    if (d.width % 256 == 0 && d.mlp_dim % 256 == 0) {
        const size_t wq = (size_t)3 * d.width * d.width * 2, w1 = (size_t)d.mlp_dim * d.width * 2;
        const size_t vq = scd_align((size_t)3 * d.width * 4), v1 = scd_align((size_t)d.mlp_dim * 4);
        const size_t per_layer = scd_align(wq) + scd_align(w1) + 2 * vq + 2 * v1;
        SCD_HIP(hipMalloc(&e->folded_mem, per_layer * d.layers));
        char* base = (char*)e->folded_mem;
        for (int l = 0; l < d.layers; ++l) {
            const void* const* lw = &e->w[W_LAYER0 + l * W_PER_LAYER];
            char* p = base + per_layer * l;
            half_t* fwq = (half_t*)p; p += scd_align(wq);
            half_t* fw1 = (half_t*)p; p += scd_align(w1);
            float* csq = (float*)p; p += vq;
            float* bq = (float*)p; p += vq;
            float* cs1 = (float*)p; p += v1;
            float* b1 = (float*)p;
            fold_ln_kernel<<<(3 * d.width + 3) / 4, 256>>>((const half_t*)lw[L_QKV_W], (const float*)lw[L_LN1_W], (const float*)lw[L_LN1_B],
                                                           (const float*)lw[L_QKV_B], 3 * d.width, d.width, fwq, csq, bq);
            fold_ln_kernel<<<(d.mlp_dim + 3) / 4, 256>>>((const half_t*)lw[L_FC1_W], (const float*)lw[L_LN2_W], (const float*)lw[L_LN2_B],
                                                         (const float*)lw[L_FC1_B], d.mlp_dim, d.width, fw1, cs1, b1);
            e->folded.push_back({fwq, fw1, csq, bq, cs1, b1});
        }
        SCD_HIP(hipDeviceSynchronize());
    }
    *out = e;
    return SCD_OK;
}

extern "C" int scd_encoder_destroy(scd_encoder* e) {
    if (e && e->folded_mem) hipFree(e->folded_mem);
    delete e;
    return SCD_OK;
}

extern "C" int scd_encoder_timing(scd_encoder* e, int enable, double* ms_out, int* launches_out, double* flop_out) {
    SCD_REQUIRE(e, "scd_encoder_timing: null encoder");
    double ms = 0.0;
    for (auto& pr : e->ev) {
        SCD_HIP(hipEventSynchronize(pr.second));
        float t = 0.f;
        SCD_HIP(hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        hipEventDestroy(pr.first);
        hipEventDestroy(pr.second);
    }
    if (ms_out) *ms_out = ms;
    if (launches_out) *launches_out = (int)e->ev.size();
    if (flop_out) *flop_out = e->timed_flop;
    e->ev.clear();
    e->timed_flop = 0.0;
    e->timing = enable != 0;
    return SCD_OK;
}

extern "C" size_t scd_encoder_ws_bytes(const scd_encoder* e, int batch) {
    if (!e || batch <= 0) return 0;
    return carve(e->d, make_pad(e->d, batch), nullptr).total;
}

static int attn_xmode() {
    static const int x = SCD_ABLATE_ENV("SCD_ATTN_X", 0);   // timing ablations only
    return x;
}

// LayerNorm folded into the QKV / fc1 GEMMs?  (then the kernel that writes the first block's input also writes its row statistics)
static bool ln_fused(const scd_encoder* e, const EncPad& pad) {
    static const int ln_fuse_env = getenv("SCD_LN_FUSE") ? atoi(getenv("SCD_LN_FUSE")) : 1;
    return ln_fuse_env && !e->folded.empty() && pad.rows % 256 == 0;
}
static int run_blocks(const scd_encoder* e, const EncWs& w, const EncPad& pad, hipStream_t st, bool* selected_out) {
    const scd_encoder_desc& d = e->d;
    const long long rows = pad.rows;
    const int bp = pad.batch;          // attention runs over the real images only
    const int act = d.act == 0 ? SCD_ACT_QUICKGELU : SCD_ACT_GELU;
    const int causal = d.kind == 1;
    // LayerNorm folded into the GEMMs (gemm.h scd_gemm_ln): the QKV / fc1 GEMMs read the raw residual stream x and apply
    // mean / rstd in their epilogue; the row statistics come from the epilogue of the GEMM that wrote x (proj / fc2), for
    // layer 0 from the kernel that writes x (assemble_visual_kernel / embed_text_kernel).  Two of the seven kernels of a block disappear.  SCD_LN_FUSE=0 restores the LN kernels.
    const bool fuse = ln_fused(e, pad);
    // Only the CLS (EOT) row of the last block's output is ever used (LN_post -> projection): after its attention the last
    // block continues on those rows alone - the output projection, LayerNorm, fc1 and fc2 of the other 196 (76) tokens of
    // every image are never computed.  Identical features (each row's arithmetic is unchanged).  SCD_LAST_SEL=0 disables it.
    static const int last_sel_env = getenv("SCD_LAST_SEL") ? atoi(getenv("SCD_LAST_SEL")) : 1;
    const bool last_sel = last_sel_env && d.mlp_dim % 256 == 0 && d.width % 256 == 0;
    *selected_out = last_sel;
    // (layer 0's statistics of x were written by assemble_visual_kernel / embed_text_kernel together with x)
    for (int l = 0; l < d.layers; ++l) {
        const void* const* lw = &e->w[W_LAYER0 + l * W_PER_LAYER];
        int rc;
        // last block, fused path: only the CLS / EOT query is needed, so K and V are projected for all rows (two thirds of
        // the QKV GEMM), Q for the selected rows only, and the attention is one query per (image, head)
        static const int last_q_env = getenv("SCD_LAST_Q") ? atoi(getenv("SCD_LAST_Q")) : 1;
        const bool last_q = last_sel && fuse && last_q_env && l == d.layers - 1 && d.width % 128 == 0 && pad.tokens <= 256;
        if (last_q) {
            const int bh = pad.bh;
            const size_t wk = (size_t)d.width * d.width;
            rc = scd_gemm_ln_finish(w.stats_a, rows, 1.0f / (float)d.width, d.ln_eps, w.rs, nullptr, st);
            if (rc) return rc;
            scd_gemm_ln lkv{w.stats_a, e->folded[l].csq + d.width, 1.0f / (float)d.width, d.ln_eps, nullptr, nullptr, w.rs};
            rc = scd_gemm_launch_ln(w.x, e->folded[l].wq + wk, e->folded[l].bq + d.width, nullptr, w.qkv, rows, 2 * d.width, d.width,
                                    SCD_ACT_NONE, &lkv, st);
            if (rc) return rc;
            gather_rows_stats_kernel<<<(unsigned)scd_cdiv(bh, 4), 256, 0, st>>>(w.x, w.stats_a, w.rows, bh, d.width, w.xsel, w.stats_sel);
            rc = scd_gemm_ln_finish(w.stats_sel, bh, 1.0f / (float)d.width, d.ln_eps, w.rs, nullptr, st);
            if (rc) return rc;
            scd_gemm_ln lq{w.stats_sel, e->folded[l].csq, 1.0f / (float)d.width, d.ln_eps, nullptr, nullptr, w.rs};
            rc = scd_gemm_launch_ln(w.xsel, e->folded[l].wq, e->folded[l].bq, nullptr, w.hsel, bh, d.width, d.width, SCD_ACT_NONE, &lq, st);
            if (rc) return rc;
            const int items = bp * d.heads;
            attention_single_query_kernel<<<(unsigned)scd_cdiv(items, 4), 256, 0, st>>>(w.qkv, w.hsel, w.rows, w.ysel, pad.tokens, d.width,
                                                                                         d.heads, items, causal);
            SCD_HIP(hipMemsetAsync(w.stats_sel, 0, (size_t)bh * 16, st));
            scd_gemm_ln lo{nullptr, nullptr, 0.f, 0.f, w.stats_sel, nullptr, nullptr};
            rc = scd_gemm_launch_ln(w.ysel, (const half_t*)lw[L_PROJ_W], (const float*)lw[L_PROJ_B], w.xsel, w.xsel, bh, d.width, d.width,
                                    SCD_ACT_NONE, &lo, st);
            if (rc) return rc;
            rc = scd_gemm_ln_finish(w.stats_sel, bh, 1.0f / (float)d.width, d.ln_eps, w.rs, nullptr, st);
            if (rc) return rc;
            scd_gemm_ln li{w.stats_sel, e->folded[l].cs1, 1.0f / (float)d.width, d.ln_eps, nullptr, nullptr, w.rs};
            rc = scd_gemm_launch_ln(w.xsel, e->folded[l].w1, e->folded[l].b1, nullptr, w.hsel, bh, d.mlp_dim, d.width, act, &li, st);
            if (rc) return rc;
            rc = scd_gemm_launch(w.hsel, (const half_t*)lw[L_FC2_W], (const float*)lw[L_FC2_B], w.xsel, w.xsel, bh, d.width, d.mlp_dim,
                                 SCD_ACT_NONE, st);
            if (rc) return rc;
            break;
        }
        if (fuse) {
            rc = scd_gemm_ln_finish(w.stats_a, rows, 1.0f / (float)d.width, d.ln_eps, w.rs, w.stats_b, st);   // also clears stats_b
            if (rc) return rc;
            scd_gemm_ln ln{w.stats_a, e->folded[l].csq, 1.0f / (float)d.width, d.ln_eps, nullptr, w.stats_b, w.rs};
            rc = scd_gemm_launch_ln(w.x, e->folded[l].wq, e->folded[l].bq, nullptr, w.qkv, rows, 3 * d.width, d.width, SCD_ACT_NONE, &ln, st);
        } else {
            layernorm_kernel<<<(unsigned)scd_cdiv(rows, 4), 256, 0, st>>>(w.x, nullptr, rows, d.width, d.ln_eps, (const float*)lw[L_LN1_W],
                                                                           (const float*)lw[L_LN1_B], w.y);
            rc = scd_gemm_launch(w.y, (const half_t*)lw[L_QKV_W], (const float*)lw[L_QKV_B], nullptr, w.qkv, rows, 3 * d.width,
                                 d.width, SCD_ACT_NONE, st);
        }
        if (rc) return rc;
        static const int attn_persist = getenv("SCD_ATTN_PERSIST") ? atoi(getenv("SCD_ATTN_PERSIST")) : 1;
        if (pad.tokens > 192 && pad.tokens <= 224 && !causal && attn_persist) {
            constexpr int attn_lds = 2 * 2 * 224 * 128 + 7 * 4096;      // K/V double buffer + the consumers' output patches
            { const int rc_ = scd_set_max_lds((const void*)attention_persist_kernel<false>, attn_lds); if (rc_) return rc_; }
            { const int rc_ = scd_set_max_lds((const void*)attention_persist_kernel<true>, attn_lds); if (rc_) return rc_; }
            const int items = bp * d.heads;
            static const int attn_t197 = getenv("SCD_ATTN_T197") ? atoi(getenv("SCD_ATTN_T197")) : 1;   // 0: the generic kernel at T = 197 as well (A/B)
            if (pad.tokens == 197 && attn_t197)
                attention_persist_kernel<true><<<items < 256 ? items : 256, 512, attn_lds, st>>>(w.qkv, w.y, pad.tokens, d.width, d.heads, items, attn_xmode());
            else
                attention_persist_kernel<false><<<items < 256 ? items : 256, 512, attn_lds, st>>>(w.qkv, w.y, pad.tokens, d.width, d.heads, items, attn_xmode());
        } else if (pad.tokens == 197) attention_kernel<7><<<bp * d.heads, 256, 0, st>>>(w.qkv, w.y, pad.tokens, d.width, d.heads, causal, attn_xmode());
        else if (pad.tokens <= 32) {
            static const int attn_short = getenv("SCD_ATTN_SHORT") ? atoi(getenv("SCD_ATTN_SHORT")) : 1;      // 0: a block per item (attention_kernel<1>; A/B, same bits)
            if (attn_short) attention_short_kernel<<<(bp * d.heads + 3) / 4, 256, 0, st>>>(w.qkv, w.y, pad.tokens, d.width, d.heads, causal, bp * d.heads);
            else attention_kernel<1><<<bp * d.heads, 256, 0, st>>>(w.qkv, w.y, pad.tokens, d.width, d.heads, causal, attn_xmode());
        }
        else if (pad.tokens <= 64) attention_kernel<2><<<bp * d.heads, 256, 0, st>>>(w.qkv, w.y, pad.tokens, d.width, d.heads, causal, attn_xmode());
        else attention_kernel<3><<<bp * d.heads, 256, 0, st>>>(w.qkv, w.y, pad.tokens, d.width, d.heads, causal, attn_xmode());
        if (last_sel && l == d.layers - 1) {
            const int bh = pad.bh;
            gather2_rows_kernel<<<(unsigned)scd_cdiv(bh, 4), 256, 0, st>>>(w.y, w.x, w.rows, bh, d.width, w.ysel, w.xsel);
            if (fuse) {
                SCD_HIP(hipMemsetAsync(w.stats_sel, 0, (size_t)bh * 16, st));
                scd_gemm_ln lo{nullptr, nullptr, 0.f, 0.f, w.stats_sel, nullptr, nullptr};
                rc = scd_gemm_launch_ln(w.ysel, (const half_t*)lw[L_PROJ_W], (const float*)lw[L_PROJ_B], w.xsel, w.xsel, bh, d.width, d.width,
                                        SCD_ACT_NONE, &lo, st);
                if (rc) return rc;
                rc = scd_gemm_ln_finish(w.stats_sel, bh, 1.0f / (float)d.width, d.ln_eps, w.rs, nullptr, st);
                if (rc) return rc;
                scd_gemm_ln li{w.stats_sel, e->folded[l].cs1, 1.0f / (float)d.width, d.ln_eps, nullptr, nullptr, w.rs};
                rc = scd_gemm_launch_ln(w.xsel, e->folded[l].w1, e->folded[l].b1, nullptr, w.hsel, bh, d.mlp_dim, d.width, act, &li, st);
            } else {
                rc = scd_gemm_launch(w.ysel, (const half_t*)lw[L_PROJ_W], (const float*)lw[L_PROJ_B], w.xsel, w.xsel, bh, d.width, d.width,
                                     SCD_ACT_NONE, st);
                if (rc) return rc;
                layernorm_kernel<<<(unsigned)scd_cdiv(bh, 4), 256, 0, st>>>(w.xsel, nullptr, bh, d.width, d.ln_eps, (const float*)lw[L_LN2_W],
                                                                             (const float*)lw[L_LN2_B], w.ysel);
                rc = scd_gemm_launch(w.ysel, (const half_t*)lw[L_FC1_W], (const float*)lw[L_FC1_B], nullptr, w.hsel, bh, d.mlp_dim, d.width, act, st);
            }
            if (rc) return rc;
            rc = scd_gemm_launch(w.hsel, (const half_t*)lw[L_FC2_W], (const float*)lw[L_FC2_B], w.xsel, w.xsel, bh, d.width, d.mlp_dim,
                                 SCD_ACT_NONE, st);
            if (rc) return rc;
            break;
        }
        if (fuse) {
            scd_gemm_ln ln{nullptr, nullptr, 0.f, 0.f, w.stats_b, nullptr, nullptr};
            rc = scd_gemm_launch_ln(w.y, (const half_t*)lw[L_PROJ_W], (const float*)lw[L_PROJ_B], w.x, w.x, rows, d.width, d.width,
                                    SCD_ACT_NONE, &ln, st);
            if (rc) return rc;
        } else {
            rc = scd_gemm_launch(w.y, (const half_t*)lw[L_PROJ_W], (const float*)lw[L_PROJ_B], w.x, w.x, rows, d.width, d.width,
                                 SCD_ACT_NONE, st);
            if (rc) return rc;
            layernorm_kernel<<<(unsigned)scd_cdiv(rows, 4), 256, 0, st>>>(w.x, nullptr, rows, d.width, d.ln_eps, (const float*)lw[L_LN2_W],
                                                                           (const float*)lw[L_LN2_B], w.y);
        }
        if (fuse) {   // the row pairs of LN2 (and stats_a cleared for fc2's epilogue) in front of the timed fc1 launch
            rc = scd_gemm_ln_finish(w.stats_b, rows, 1.0f / (float)d.width, d.ln_eps, w.rs, w.stats_a, st);
            if (rc) return rc;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (e->timing) {
            SCD_HIP(hipEventCreate(&e0));
            SCD_HIP(hipEventCreate(&e1));
            SCD_HIP(hipEventRecord(e0, st));
        }
        if (fuse) {
            scd_gemm_ln ln{w.stats_b, e->folded[l].cs1, 1.0f / (float)d.width, d.ln_eps, nullptr, w.stats_a, w.rs};
            rc = scd_gemm_launch_ln(w.x, e->folded[l].w1, e->folded[l].b1, nullptr, w.h, rows, d.mlp_dim, d.width, act, &ln, st);
        } else {
            rc = scd_gemm_launch(w.y, (const half_t*)lw[L_FC1_W], (const float*)lw[L_FC1_B], nullptr, w.h, rows, d.mlp_dim, d.width, act, st);
        }
        if (rc) return rc;
        if (e->timing) {
            SCD_HIP(hipEventRecord(e1, st));
            scd_encoder* me = const_cast<scd_encoder*>(e);
            me->ev.emplace_back(e0, e1);
            me->timed_flop += 2.0 * (double)rows * d.mlp_dim * d.width;
        }
        if (fuse) {
            scd_gemm_ln ln{nullptr, nullptr, 0.f, 0.f, w.stats_a, nullptr, nullptr};
            rc = scd_gemm_launch_ln(w.h, (const half_t*)lw[L_FC2_W], (const float*)lw[L_FC2_B], w.x, w.x, rows, d.width, d.mlp_dim,
                                    SCD_ACT_NONE, &ln, st);
        } else {
            rc = scd_gemm_launch(w.h, (const half_t*)lw[L_FC2_W], (const float*)lw[L_FC2_B], w.x, w.x, rows, d.width, d.mlp_dim,
                                 SCD_ACT_NONE, st);
        }
        if (rc) return rc;
    }
    return SCD_OK;
}

static int run_head(const scd_encoder* e, const EncWs& w, const EncPad& pad, bool selected, void* out, int normalize, hipStream_t st) {
    const scd_encoder_desc& d = e->d;
    const int batch = pad.batch, bp = pad.bh;
    // final LayerNorm on the CLS / EOT rows (already gathered by the last block, or gathered here), then the projection
    layernorm_kernel<<<(unsigned)scd_cdiv(bp, 4), 256, 0, st>>>(selected ? w.xsel : w.x, selected ? nullptr : w.rows, bp, d.width, d.ln_eps,
                                                                 (const float*)e->w[W_LNPOST_W], (const float*)e->w[W_LNPOST_B], w.cls);
    const half_t* fin = w.cls;
    int dim = d.width;
    if (d.out_dim > 0) {
        int rc = scd_gemm_launch(w.cls, (const half_t*)e->w[W_PROJ], nullptr, nullptr, w.outp, bp, d.out_dim, d.width, SCD_ACT_NONE, st);
        if (rc) return rc;
        fin = w.outp;
        dim = d.out_dim;
    }
    emit_kernel<<<(unsigned)scd_cdiv(batch, 4), 256, 0, st>>>(fin, batch, dim, normalize, (half_t*)out);
    SCD_LAUNCH_CHECK();
    return SCD_OK;
}

extern "C" int scd_vit_encode_image(scd_handle h, const scd_encoder* e, const void* pixels, int dtype, int batch, void* out,
                                    int normalize, void* ws, size_t ws_bytes, void* stream_) {
    SCD_REQUIRE(h && e && pixels && out && ws && batch > 0, "scd_vit_encode_image: bad arguments");
    { const int rc_ = scd_check_device(h, "scd_vit_encode_image"); if (rc_) return rc_; }
    SCD_REQUIRE(e->d.kind == 0 || e->d.kind == 2, "scd_vit_encode_image: encoder is not a visual tower");
    SCD_REQUIRE(dtype == SCD_F32 || dtype == SCD_F16, "scd_vit_encode_image: bad dtype %d", dtype);
    SCD_REQUIRE(ws_bytes >= scd_encoder_ws_bytes(e, batch), "scd_vit_encode_image: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const scd_encoder_desc& d = e->d;
    const EncPad pad = make_pad(d, batch);
    EncWs w = carve(d, pad, (char*)ws);
    const int kk = 3 * d.patch * d.patch;
    const long long total8 = pad.prows * kk / 8;
    half_t* cols = w.qkv;      // [prows, kk] scratch (fits: kk <= 3*width, prows <= rows)
    SCD_REQUIRE(kk <= 3 * d.width, "scd_vit_encode_image: patch too large for scratch");
    int rc;
    // fp16 pixels, 16 x 16 patches, a row count the four-wave GEMM takes: the patch GEMM gathers its operand from the image (round 6);
    // anything else goes through the im2col matrix as before (same products in the same order: the features do not change)
    static const bool img_env = !(getenv("SCD_PATCH_FROM_IMAGE") && atoi(getenv("SCD_PATCH_FROM_IMAGE")) == 0);   // =0: the im2col path (A/B; same bits)
    const bool from_image = img_env && dtype == SCD_F16 && d.patch == 16 && d.image % 16 == 0 && pad.prows % 256 == 0 && d.width % 256 == 0 &&
                            (double)batch * 3.0 * d.image * d.image * 2.0 < 4294967296.0;
    if (from_image) {
        rc = scd_gemm_launch_img((const half_t*)pixels, (const half_t*)e->w[W_PATCH], w.y, pad.prows, d.width, batch, d.image, st);
    } else {
        if (dtype == SCD_F32) im2col_kernel<float><<<(unsigned)scd_cdiv(total8, 256), 256, 0, st>>>((const float*)pixels, batch, pad.prows, d.image, d.patch, cols);
        else im2col_kernel<half_t><<<(unsigned)scd_cdiv(total8, 256), 256, 0, st>>>((const half_t*)pixels, batch, pad.prows, d.image, d.patch, cols);
        rc = scd_gemm_launch(cols, (const half_t*)e->w[W_PATCH], nullptr, nullptr, w.y, pad.prows, d.width, kk, SCD_ACT_NONE, st);
    }
    if (rc) return rc;
    const long long rows = pad.rows;
    static const int asm_rb = getenv("SCD_ASSEMBLE_ROWS") ? atoi(getenv("SCD_ASSEMBLE_ROWS")) : 4;     // 1: one row per wave (round 1-5 kernel; A/B, same bits); 8: eight
    if (asm_rb > 1 && d.width <= 1024) {
#define ASM_GO(RB)                                                                                                                        \
    do {                                                                                                                                  \
        const long long items = (long long)d.tokens * ((batch + RB - 1) / RB) + (rows - (long long)batch * d.tokens);                     \
        assemble_visual_rows_kernel<RB><<<(unsigned)scd_cdiv(items, 4), 256, 0, st>>>(w.y, (const float*)e->w[W_PATCH_B], (const float*)e->w[W_CLS], \
                                                                                      (const float*)e->w[W_POS], rows, batch, d.tokens, d.width,      \
                                                                                      (const float*)e->w[W_LNPRE_W], (const float*)e->w[W_LNPRE_B],   \
                                                                                      d.ln_eps, w.x, ln_fused(e, pad) ? w.stats_a : nullptr);          \
    } while (0)
        if (asm_rb == 8) ASM_GO(8); else ASM_GO(4);       // measured (tools/asm_rows_sweep.sh): 1: 864, 2: 600, 3: 524, 4: 504, 6: 578, 8: 566, 16: 761 us
#undef ASM_GO
    } else {
        assemble_visual_kernel<<<(unsigned)scd_cdiv(rows, 4), 256, 0, st>>>(w.y, (const float*)e->w[W_PATCH_B], (const float*)e->w[W_CLS],
                                                                             (const float*)e->w[W_POS], rows, batch, d.tokens, d.width,
                                                                             (const float*)e->w[W_LNPRE_W], (const float*)e->w[W_LNPRE_B],
                                                                             d.ln_eps, w.x, ln_fused(e, pad) ? w.stats_a : nullptr);
    }
    cls_rows_kernel<<<(pad.bh + 255) / 256, 256, 0, st>>>(w.rows, pad.bh, d.tokens, batch, 0);
    bool selected = false;
    rc = run_blocks(e, w, pad, st, &selected);
    if (rc) return rc;
    return run_head(e, w, pad, selected, out, normalize, st);
}

extern "C" int scd_clip_encode_text(scd_handle h, const scd_encoder* e, const int32_t* tokens, int batch, void* out, int normalize,
                                    void* ws, size_t ws_bytes, void* stream_) {
    SCD_DEVICE_ENTRY(h, "scd_clip_encode_text");
    SCD_REQUIRE(e, "scd_clip_encode_text: null encoder");
    return scd_clip_encode_text_len(h, e, tokens, batch, e->d.tokens, out, normalize, ws, ws_bytes, stream_);
}

// The text tower is causal and only the EOT position's output is used (model.py encode_text: x[arange, text.argmax(-1)]), so the
// positions behind the last EOT of the batch never influence a result: with ctx_len >= 1 + max_b argmax_t tokens[b][t] only the
// first ctx_len positions are computed.  Prompts are ~10-20 tokens of the 77: 4-5x fewer token rows through every GEMM, the same
// bits out (masked keys contribute exact zeros to the softmax sums and to P V, and the GEMM rows are independent of M).
extern "C" int scd_clip_encode_text_len(scd_handle h, const scd_encoder* e, const int32_t* tokens, int batch, int ctx_len, void* out,
                                        int normalize, void* ws, size_t ws_bytes, void* stream_) {
    SCD_REQUIRE(h && e && tokens && out && ws && batch > 0, "scd_clip_encode_text: bad arguments");
    { const int rc_ = scd_check_device(h, "scd_clip_encode_text"); if (rc_) return rc_; }
    SCD_REQUIRE(e->d.kind == 1, "scd_clip_encode_text: encoder is not a text tower");
    SCD_REQUIRE(ctx_len >= 1 && ctx_len <= e->d.tokens, "scd_clip_encode_text_len: ctx_len=%d must be in [1, %d]", ctx_len, e->d.tokens);
    SCD_REQUIRE(ws_bytes >= scd_encoder_ws_bytes(e, batch), "scd_clip_encode_text: workspace too small");
    hipStream_t st = (hipStream_t)stream_;
    const scd_encoder_desc& d = e->d;
    const EncPad pad = make_pad(d, batch, ctx_len);
    EncWs w = carve(d, pad, (char*)ws);
    const long long rows = pad.rows;
    embed_text_kernel<<<(unsigned)scd_cdiv(rows, 4), 256, 0, st>>>(tokens, batch, (const half_t*)e->w[W_PATCH], d.vocab,
                                                                    (const float*)e->w[W_POS], rows, pad.tokens, d.tokens, d.width, w.x, w.rows,
                                                                    ln_fused(e, pad) ? w.stats_a : nullptr);
    cls_rows_kernel<<<(pad.bh + 255) / 256, 256, 0, st>>>(w.rows, pad.bh, pad.tokens, batch, 1);
    bool selected = false;
    int rc = run_blocks(e, w, pad, st, &selected);
    if (rc) return rc;
    return run_head(e, w, pad, selected, out, normalize, st);
}
