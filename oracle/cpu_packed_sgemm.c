/* TEST / MEASUREMENT INFRASTRUCTURE ONLY -- the GEMM of bench.py's `cpu_baseline` leg.
 *
 * The reference's default sgemm is the un-vendored crate `matrixmultiply` 0.3.10 (Cargo.toml:16, Cargo.lock),
 * called single-threaded (no `threading` feature) from src/gemm.rs:102-117 with element strides
 * (rsa, csa, rsb, csb, rsc = n, csc = 1) derived from the transposition flags (gemm.rs:88-98).  Its published
 * algorithm is the BLIS / Goto loop nest: C is cut into nc-wide column panels and kc-deep k slabs; the kc x nc slab
 * of B is PACKED into NR-wide strips, each mc x kc block of A is PACKED into MR-wide strips (packing absorbs any
 * strides, so the transposed layouts cost nothing extra), and an MR x NR register-blocked FMA micro-kernel walks the
 * packed strips; C = beta*C + alpha*A*B with beta applied on the first k slab only.  This file restates that
 * algorithm in portable C with GCC vector extensions (one vector = 8 floats on AVX, 16 on AVX-512; built
 * -O3 -march=native on the box it is timed on): a 6 x 2-vector micro-tile on AVX (12 accumulators, the shape BLIS
 * uses on Haswell-class cores -- matrixmultiply's own AVX/FMA sgemm tile is 8 x 8), 12 x 2 vectors on AVX-512.
 *
 * The PARITY oracle keeps its plain k-ordered triple loop (taper_oracle.c: ot_sgemm_rowmajor); this kernel is linked
 * in only when the library is built with -DOT_PACKED_SGEMM (`make fast`), and tests/test_cpu_baseline.py checks it
 * against the plain loop on every layout / ragged shape. */
#include <stdlib.h>
#include <string.h>

#if defined(__AVX512F__)
#define VW 16
#define MR 12
#else
#define VW 8
#define MR 6
#endif
#define NR (2 * VW)
#define KC 256
#define MC (MR * 12)
#define NC (NR * 128)

typedef float vf __attribute__((vector_size(VW * 4), aligned(4)));   /* unaligned loads / stores */

static inline vf bcast(float x) { return x + (vf){0.0f}; }   /* scalar (op) vector broadcasts the scalar */

/* pack an (mc x kc) block of op(A) into MR-row strips: strip s holds, for every p, the MR values A[s*MR + r][p] */
static void pack_a(float *restrict dst, const float *a, long rs, long cs, int mc, int kc) {
    for (int i0 = 0; i0 < mc; i0 += MR) {
        const int mr = mc - i0 < MR ? mc - i0 : MR;
        for (int p = 0; p < kc; ++p) {
            int r = 0;
            for (; r < mr; ++r) dst[r] = a[(long)(i0 + r) * rs + (long)p * cs];
            for (; r < MR; ++r) dst[r] = 0.0f;
            dst += MR;
        }
    }
}

/* pack a (kc x nc) slab of op(B) into NR-column strips */
static void pack_b(float *restrict dst, const float *b, long rs, long cs, int kc, int nc) {
    for (int j0 = 0; j0 < nc; j0 += NR) {
        const int nr = nc - j0 < NR ? nc - j0 : NR;
        if (cs == 1 && nr == NR) {
            for (int p = 0; p < kc; ++p) {
                memcpy(dst, b + (long)p * rs + j0, NR * sizeof(float));
                dst += NR;
            }
        } else {
            for (int p = 0; p < kc; ++p) {
                int c = 0;
                for (; c < nr; ++c) dst[c] = b[(long)p * rs + (long)(j0 + c) * cs];
                for (; c < NR; ++c) dst[c] = 0.0f;
                dst += NR;
            }
        }
    }
}

/* acc[MR][2] = sum_p a_strip[p][:] (x) b_strip[p][:] */
static inline void micro_kernel(int kc, const float *restrict ap, const float *restrict bp, vf acc[MR][2]) {
    vf c[MR][2];
    for (int r = 0; r < MR; ++r) c[r][0] = c[r][1] = bcast(0.0f);
    for (int p = 0; p < kc; ++p) {
        const vf b0 = *(const vf *)(bp), b1 = *(const vf *)(bp + VW);
#pragma GCC unroll 12
        for (int r = 0; r < MR; ++r) {
            const float a = ap[r];
            c[r][0] += a * b0;
            c[r][1] += a * b1;
        }
        ap += MR;
        bp += NR;
    }
    for (int r = 0; r < MR; ++r) {
        acc[r][0] = c[r][0];
        acc[r][1] = c[r][1];
    }
}

/* gemm.rs:72-119 argument semantics (row-major C with ldc = n) */
void ot_packed_sgemm_rowmajor(int trans_a, int trans_b, int m, int n, int k, float alpha, const float *a, const float *b, float beta,
                              float *c) {
    const long a_rs = trans_a ? 1 : k, a_cs = trans_a ? m : 1;   /* gemm.rs:88-93 */
    const long b_rs = trans_b ? 1 : n, b_cs = trans_b ? k : 1;   /* gemm.rs:94-98 */
    if (m <= 0 || n <= 0) return;
    if (k <= 0 || alpha == 0.0f) {
        for (long i = 0; i < (long)m * n; ++i) c[i] = beta == 0.0f ? 0.0f : beta * c[i];
        return;
    }
    float *apack = NULL, *bpack = NULL;
    if (posix_memalign((void **)&apack, 64, (size_t)MC * KC * sizeof(float)) ||
        posix_memalign((void **)&bpack, 64, (size_t)KC * (NC < ((n + NR - 1) / NR) * NR ? NC : ((n + NR - 1) / NR) * NR) * sizeof(float)))
        abort();
    for (int jc = 0; jc < n; jc += NC) {
        const int nc = n - jc < NC ? n - jc : NC;
        for (int pc = 0; pc < k; pc += KC) {
            const int kc = k - pc < KC ? k - pc : KC;
            const float beta_eff = pc == 0 ? beta : 1.0f;
            pack_b(bpack, b + (long)pc * b_rs + (long)jc * b_cs, b_rs, b_cs, kc, nc);
            for (int ic = 0; ic < m; ic += MC) {
                const int mc = m - ic < MC ? m - ic : MC;
                pack_a(apack, a + (long)ic * a_rs + (long)pc * a_cs, a_rs, a_cs, mc, kc);
                for (int jr = 0; jr < nc; jr += NR) {
                    const int nr = nc - jr < NR ? nc - jr : NR;
                    const float *bp = bpack + (long)(jr / NR) * kc * NR;
                    for (int ir = 0; ir < mc; ir += MR) {
                        const int mr = mc - ir < MR ? mc - ir : MR;
                        vf acc[MR][2];
                        micro_kernel(kc, apack + (long)(ir / MR) * kc * MR, bp, acc);
                        float *ct = c + (long)(ic + ir) * n + jc + jr;
                        if (nr == NR) {
                            const vf va = bcast(alpha), vb = bcast(beta_eff);
                            for (int r = 0; r < mr; ++r) {
                                vf *c0 = (vf *)(ct + (long)r * n), *c1 = (vf *)(ct + (long)r * n + VW);
                                if (beta_eff == 0.0f) {
                                    *c0 = va * acc[r][0];
                                    *c1 = va * acc[r][1];
                                } else {
                                    *c0 = vb * *c0 + va * acc[r][0];
                                    *c1 = vb * *c1 + va * acc[r][1];
                                }
                            }
                        } else {
                            for (int r = 0; r < mr; ++r)
                                for (int q = 0; q < nr; ++q) {
                                    const float v = alpha * acc[r][q / VW][q % VW];
                                    float *dst = ct + (long)r * n + q;
                                    *dst = beta_eff == 0.0f ? v : beta_eff * *dst + v;
                                }
                        }
                    }
                }
            }
        }
    }
    free(apack);
    free(bpack);
}
