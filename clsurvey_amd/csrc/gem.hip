// GEM memory-gradient kernels — restates methods/rehearsal/model/gem.py:20-80 (store_grad,
// overwrite_grad, project2cone2) and :275-277 (violation test).
//
// The reference keeps grads as a [P][n_tasks] matrix (column per task), copies it to the CPU and runs
// numpy float64 GEMMs over P for the QP inputs (gem.py:58-76).  Here a task's gradient is one
// CONTIGUOUS row of G[n_tasks][ld] (the ParamArena's gradient is already flat, so store_grad is a
// single axpy/copy), and ONE pass over the m selected rows yields the whole m x m Gram matrix in f64:
// 4*m bytes per parameter instead of the 4*(t+1) + 8*t*... of the numpy path.  Only the m*m doubles
// travel to the host for the tiny QP.
#include "common.hpp"
#include <utility>

namespace {

constexpr int GB = 256;
constexpr int MAXM = 16;

struct RowSel { int idx[MAXM]; };
struct Coefs { float v[MAXM]; };

// VEC: 16-byte accesses (the arena gradient and the rows of G are 16-byte aligned, ld is a multiple of 4); the tail
// n % 4 is done by the first threads.  Per-element arithmetic is the same either way.
template <bool VEC>
__global__ __launch_bounds__(GB) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n, float alpha,
                                                  int assign) {
    const size_t stride = (size_t)gridDim.x * GB, t0 = (size_t)blockIdx.x * GB + threadIdx.x;
    const size_t n4 = VEC ? n / 4 : 0;
    for (size_t i = t0; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        float4 b = assign ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(y)[i];
        b.x = assign ? alpha * a.x : b.x + alpha * a.x; b.y = assign ? alpha * a.y : b.y + alpha * a.y;
        b.z = assign ? alpha * a.z : b.z + alpha * a.z; b.w = assign ? alpha * a.w : b.w + alpha * a.w;
        reinterpret_cast<float4*>(y)[i] = b;
    }
    for (size_t i = n4 * 4 + t0; i < n; i += stride)
        y[i] = assign ? alpha * x[i] : y[i] + alpha * x[i];
}

// partial[b][pair] = sum over this block's column range of G[ri][c] * G[rj][c]   (i <= j), pair = the index in row-major
// order of the upper triangle.
//
// M(M+1)/2 f64 accumulators per thread do not fit the register file beyond M = 6 (M = 16: 136 doubles = 272 registers; the
// round-3 kernel spilled 29 .. 205 scratch instructions for M = 7 .. 16, and config 4 with 9 tasks in memory runs M = 10).
// With the row indices of every pair as compile-time constants (GramPair) the accumulators of up to 11 rows (66 doubles) stay
// in registers.  Beyond that the pairs are dealt out over S = 2 (16 rows: 4) of the block's WAVES (wave w holds the pairs p = w % S,
// w % S + S, ...; uniform per wave, no divergence): the waves of a set read the same 64 column groups (the second read of
// a line hits the cache; HBM still sees ONE pass over G, but the load path carries the lines twice: measured 3.9 TB/s at 11
// rows with S = 2 against 5.3 at 5 rows with S = 1, which is why S = 1 is kept as far as the registers reach).
__host__ __device__ constexpr int gram_pair_i(int M, int p) {
    int i = 0;
    while (p >= M - i) { p -= M - i; ++i; }
    return i;
}
__host__ __device__ constexpr int gram_pair_j(int M, int p) {
    int i = 0;
    while (p >= M - i) { p -= M - i; ++i; }
    return i + p;
}
__host__ __device__ constexpr int gram_split(int M) { return M <= 11 ? 1 : M <= 15 ? 2 : 4; }

// pair K of wave group GRP: compile-time row indices (as template constants: a loop variable through a constexpr function
// leaves the row array indexed at run time, i.e. in scratch)
template <int M, int S, int GRP, int K>
struct GramPair {
    static constexpr int p = GRP + K * S, i = gram_pair_i(M, p), j = gram_pair_j(M, p);
};
template <int M, int S, int GRP, int... K>
__device__ __forceinline__ void gram_acc4(double* acc, const float4* v, std::integer_sequence<int, K...>) {
    ((acc[K] += (double)v[GramPair<M, S, GRP, K>::i].x * (double)v[GramPair<M, S, GRP, K>::j].x,
      acc[K] += (double)v[GramPair<M, S, GRP, K>::i].y * (double)v[GramPair<M, S, GRP, K>::j].y,
      acc[K] += (double)v[GramPair<M, S, GRP, K>::i].z * (double)v[GramPair<M, S, GRP, K>::j].z,
      acc[K] += (double)v[GramPair<M, S, GRP, K>::i].w * (double)v[GramPair<M, S, GRP, K>::j].w), ...);
}
template <int M, int S, int GRP, int... K>
__device__ __forceinline__ void gram_acc1(double* acc, const float* v, std::integer_sequence<int, K...>) {
    ((acc[K] += (double)v[GramPair<M, S, GRP, K>::i] * (double)v[GramPair<M, S, GRP, K>::j]), ...);
}

// Accumulation of wave group GRP's pairs over this block's columns; acc is padded to the same NA_MAX slots in every group
// (slots past the group's own pair count stay 0) so that the reduction below is ONE piece of code for the whole block.
template <int M, int S, int GRP>
__device__ __forceinline__ void gram_accumulate(const float* __restrict__ G, size_t ld, const RowSel& sel, size_t n, bool vec,
                                                double (&acc)[(M * (M + 1) / 2 + S - 1) / S]) {
    constexpr int NP = M * (M + 1) / 2;
    constexpr int NA = (NP - GRP + S - 1) / S;            // pairs GRP, GRP + S, ... of this wave
    constexpr int NA_MAX = (NP + S - 1) / S;
    constexpr int COLS = GB / S;                          // column groups per block and sweep
#pragma unroll
    for (int k = 0; k < NA_MAX; ++k) acc[k] = 0.0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t col0 = (size_t)blockIdx.x * COLS + (size_t)(wave / S) * 64 + lane, stride = (size_t)gridDim.x * COLS;
    const size_t n4 = vec ? n / 4 : 0;      // (a run-time flag: the 16-byte loop and the scalar loop are both in every instance anyway)
    for (size_t c4 = col0; c4 < n4; c4 += stride) {
        float4 v[M];
#pragma unroll
        for (int i = 0; i < M; ++i) v[i] = reinterpret_cast<const float4*>(G + (size_t)sel.idx[i] * ld)[c4];
        gram_acc4<M, S, GRP>(acc, v, std::make_integer_sequence<int, NA>{});
    }
    for (size_t c = n4 * 4 + col0; c < n; c += stride) {
        float v[M];
#pragma unroll
        for (int i = 0; i < M; ++i) v[i] = G[(size_t)sel.idx[i] * ld + c];
        gram_acc1<M, S, GRP>(acc, v, std::make_integer_sequence<int, NA>{});
    }
}

template <int M>
__global__ __launch_bounds__(GB) void gram_partial_kernel(const float* __restrict__ G, size_t ld, RowSel sel, size_t n, int vec,
                                                          double* __restrict__ partial) {
    constexpr int S = gram_split(M);
    constexpr int NP = M * (M + 1) / 2;
    constexpr int NA_MAX = (NP + S - 1) / S;
    __shared__ double red[GB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = (tid >> 6) % S;                       // uniform per wave
    double acc[NA_MAX];
    // only the ACCUMULATION is specific to the wave group (compile-time pair indices); no barrier inside the branches
    if constexpr (S == 1) {
        gram_accumulate<M, 1, 0>(G, ld, sel, n, vec != 0, acc);
    } else if constexpr (S == 2) {
        if (grp) gram_accumulate<M, 2, 1>(G, ld, sel, n, vec != 0, acc);
        else gram_accumulate<M, 2, 0>(G, ld, sel, n, vec != 0, acc);
    } else {
        if (grp == 0) gram_accumulate<M, 4, 0>(G, ld, sel, n, vec != 0, acc);
        else if (grp == 1) gram_accumulate<M, 4, 1>(G, ld, sel, n, vec != 0, acc);
        else if (grp == 2) gram_accumulate<M, 4, 2>(G, ld, sel, n, vec != 0, acc);
        else gram_accumulate<M, 4, 3>(G, ld, sel, n, vec != 0, acc);
    }
    // common code, every wave at the same barriers: fixed tree, first over the wave sets (offsets that keep wave % S), then
    // inside the wave; one pair slot per round (round k: pair grp + k * S of each wave group)
#pragma unroll
    for (int k = 0; k < NA_MAX; ++k) {
        red[tid] = acc[k];
        __syncthreads();
        for (int o = GB / 2; o >= 64 * S; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        for (int o = 32; o > 0; o >>= 1) {
            if (tid < 64 * S && lane < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid < 64 * S && lane == 0 && grp + k * S < NP) partial[(size_t)blockIdx.x * NP + grp + k * S] = red[tid];
        __syncthreads();
    }
}

// One block per pair (i <= j): 256 threads sum the per-block partials b = t, t + 256, ... and a fixed tree adds the 256
// sums (a single thread walking 2048 partials is 2048 dependent memory round trips: it took longer than the pass over G).
__global__ __launch_bounds__(256) void gram_final_kernel(const double* __restrict__ partial, int nblocks, int M,
                                                         double* __restrict__ out) {
    __shared__ double red[256];
    const int NP = M * (M + 1) / 2;
    const int p = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(size_t)b * NP + p];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int i = 0, rem = p;                       // unpack pair index -> (i, j)
        while (rem >= M - i) { rem -= M - i; ++i; }
        const int j = i + rem;
        out[i * M + j] = red[0];
        out[j * M + i] = red[0];
    }
}

template <bool VEC>
__global__ __launch_bounds__(GB) void project_kernel(const float* __restrict__ G, size_t ld, RowSel sel, Coefs cf, int m,
                                                     const float* __restrict__ g, float* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * GB;
    const size_t n4 = VEC ? n / 4 : 0;
    for (size_t c4 = (size_t)blockIdx.x * GB + threadIdx.x; c4 < n4; c4 += stride) {
        const float4 gv = reinterpret_cast<const float4*>(g)[c4];
        double s0 = (double)gv.x, s1 = (double)gv.y, s2 = (double)gv.z, s3 = (double)gv.w;
        for (int i = 0; i < m; ++i) {
            const float4 r = reinterpret_cast<const float4*>(G + (size_t)sel.idx[i] * ld)[c4];
            const double vi = (double)cf.v[i];
            s0 += vi * (double)r.x; s1 += vi * (double)r.y; s2 += vi * (double)r.z; s3 += vi * (double)r.w;
        }
        reinterpret_cast<float4*>(out)[c4] = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
    }
    for (size_t c = n4 * 4 + (size_t)blockIdx.x * GB + threadIdx.x; c < n; c += stride) {
        // gem.py:79: x = v . memories + gradient, evaluated in f64 like numpy, rounded once to f32
        double s = (double)g[c];
        for (int i = 0; i < m; ++i) s += (double)cf.v[i] * (double)G[(size_t)sel.idx[i] * ld + c];
        out[c] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// GEM's QP on the device (rehearsal/model/gem.py:58-80): from the f64 Gram matrix of [memory rows..., current row]
//     P = 1/2 (M M^T + (M M^T)^T) + eps I,   q = -M g,   minimise 1/2 v^T P v - q^T v   subject to  v >= margin
// solved by the Goldfarb-Idnani dual active-set method (the algorithm of quadprog.solve_qp, which the reference calls;
// Math. Programming 27, 1983), restated for at most QP_MAX unknowns in f64 — the same steps as methods/qp.py, which stays
// as the host-side cross-check.  t <= 9 here: the whole solve is a few thousand flops, so ONE lane does it; what matters is
// that neither the Gram matrix nor v ever visits the host (the previous form synchronised the stream once per observe).
// info[0] = number of violated constraints g.G_k < 0 (gem.py:275-277: 0 => no projection), info[1] = status (0 ok).
constexpr int QP_MAX = 16;

// Every array of the solver is indexed at run time, which in registers / private memory means scratch (80 scratch
// instructions in the round-3 build): the one lane that solves keeps its matrices in LDS instead.
struct QpWork {
    double G[QP_MAX][QP_MAX], Ginv[QP_MAX][QP_MAX], L[QP_MAX][QP_MAX], Li[QP_MAX][QP_MAX], S[QP_MAX][QP_MAX], B[QP_MAX][QP_MAX];
    double a[QP_MAX], x[QP_MAX], b[QP_MAX], z[QP_MAX], r[QP_MAX], u[QP_MAX + 1];
    int active[QP_MAX];
};

__device__ void qp_inverse_spd(const double (*A)[QP_MAX], int n, double (*Inv)[QP_MAX], double (*L)[QP_MAX], double (*Li)[QP_MAX]) {
    // Cholesky A = L L^T, Linv by forward substitution, Inv = Linv^T Linv (numpy: cholesky, inv, Linv.T @ Linv)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i][j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = (i == j) ? sqrt(s) : s / L[j][j];
        }
    for (int c = 0; c < n; ++c)
        for (int i = 0; i < n; ++i) {
            if (i < c) { Li[i][c] = 0.0; continue; }
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= L[i][k] * Li[k][c];
            Li[i][c] = s / L[i][i];
        }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int k = (i > j ? i : j); k < n; ++k) s += Li[k][i] * Li[k][j];
            Inv[i][j] = s;
        }
}

// solve S X = B for SPD S (na x na) and B (na x nb) by Gaussian elimination with partial pivoting (numpy.linalg.solve)
__device__ void qp_solve(double (*S)[QP_MAX], double (*B)[QP_MAX], int na, int nb) {
    for (int c = 0; c < na; ++c) {
        int piv = c;
        for (int r = c + 1; r < na; ++r)
            if (fabs(S[r][c]) > fabs(S[piv][c])) piv = r;
        if (piv != c) {
            for (int k = 0; k < na; ++k) { const double t = S[c][k]; S[c][k] = S[piv][k]; S[piv][k] = t; }
            for (int k = 0; k < nb; ++k) { const double t = B[c][k]; B[c][k] = B[piv][k]; B[piv][k] = t; }
        }
        for (int r = c + 1; r < na; ++r) {
            const double f = S[r][c] / S[c][c];
            for (int k = c; k < na; ++k) S[r][k] -= f * S[c][k];
            for (int k = 0; k < nb; ++k) B[r][k] -= f * B[c][k];
        }
    }
    for (int r = na - 1; r >= 0; --r)
        for (int k = 0; k < nb; ++k) {
            double s = B[r][k];
            for (int c = r + 1; c < na; ++c) s -= S[r][c] * B[c][k];
            B[r][k] = s / S[r][r];
        }
}

__global__ void gem_qp_kernel(const double* __restrict__ gram, int m, double margin, double eps, double* __restrict__ v_out,
                              int* __restrict__ info) {
    __shared__ QpWork W;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int n = m - 1;                                   // unknowns = memory rows; row / column n = current gradient
    int viol = 0;
    for (int i = 0; i < n; ++i) viol += gram[(size_t)n * m + i] < 0.0;
    info[0] = viol;
    info[1] = 0;
    for (int i = 0; i < n; ++i) v_out[i] = 0.0;
    if (viol == 0) return;
    const double tol = 1e-12;
    double (*G)[QP_MAX] = W.G, (*Ginv)[QP_MAX] = W.Ginv;
    double *a = W.a, *x = W.x, *b = W.b;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) G[i][j] = 0.5 * (gram[(size_t)i * m + j] + gram[(size_t)j * m + i]) + (i == j ? eps : 0.0);
        a[i] = -gram[(size_t)i * m + n];                   // q = -M g; quadprog minimises 1/2 x^T G x - a^T x with a = q
        b[i] = margin;
    }
    qp_inverse_spd(G, n, Ginv, W.L, W.Li);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += Ginv[i][j] * a[j];
        x[i] = s;                                          // unconstrained minimum
    }
    int* active = W.active;
    int na = 0;
    double* u = W.u;
    for (int it = 0; it < 200; ++it) {
        // constraints are v_i >= margin (C = I): slack s_i = x_i - b_i; most violated inactive one
        int p = -1;
        double worst = 0.0;
        for (int i = 0; i < n; ++i) {
            bool act = false;
            for (int k = 0; k < na; ++k) act |= active[k] == i;
            const double si = x[i] - b[i];
            if (!act && si < -tol * fmax(1.0, fabs(b[i])) && (p < 0 || si < worst)) { p = i; worst = si; }
        }
        if (p < 0) {
            for (int i = 0; i < n; ++i) v_out[i] = x[i];
            return;
        }
        u[na] = 0.0;
        for (int inner = 0; inner < 200; ++inner) {
            double *z = W.z, *r = W.r;
            if (na > 0) {
                // Nstar = (N^T Ginv N)^-1 N^T Ginv ; H = Ginv - Ginv N Nstar ; z = H n_p ; r = Nstar n_p   (n_p = e_p)
                double (*S)[QP_MAX] = W.S, (*B)[QP_MAX] = W.B;
                for (int i = 0; i < na; ++i) {
                    for (int j = 0; j < na; ++j) S[i][j] = Ginv[active[i]][active[j]];
                    for (int j = 0; j < n; ++j) B[i][j] = Ginv[active[i]][j];
                }
                qp_solve(S, B, na, n);                     // B = Nstar (na x n)
                for (int j = 0; j < na; ++j) r[j] = B[j][p];
                for (int i = 0; i < n; ++i) {
                    double s = Ginv[i][p];
                    for (int k = 0; k < na; ++k) s -= Ginv[i][active[k]] * B[k][p];
                    z[i] = s;
                }
            } else {
                for (int i = 0; i < n; ++i) z[i] = Ginv[i][p];
            }
            double t1 = INFINITY, t2 = INFINITY;
            int drop = -1;
            for (int j = 0; j < na; ++j)
                if (r[j] > tol && u[j] / r[j] < t1) { t1 = u[j] / r[j]; drop = j; }
            const double zn = z[p];
            const double sp = x[p] - b[p];
            if (zn > tol) t2 = -sp / zn;
            const double t = fmin(t1, t2);
            if (!(t < INFINITY)) { info[1] = 2; return; }           // infeasible
            if (t2 < INFINITY)
                for (int i = 0; i < n; ++i) x[i] += t * z[i];
            for (int j = 0; j < na; ++j) u[j] -= t * r[j];
            u[na] += t;
            if (t == t2) { active[na++] = p; break; }                 // full step: p joins the active set
            for (int j = drop; j < na; ++j) { active[j] = active[j + 1]; u[j] = u[j + 1]; }   // partial step: drop, retry p
            --na;
            if (inner == 199) { info[1] = 1; return; }
        }
    }
    info[1] = 1;
}

template <bool VEC>
__global__ __launch_bounds__(GB) void project_dev_kernel(const float* __restrict__ G, size_t ld, RowSel sel, const double* __restrict__ v,
                                                         const int* __restrict__ info, int m, const float* __restrict__ g,
                                                         float* __restrict__ out, size_t n) {
    if (info[0] == 0 && out == g) return;                   // no violated constraint: the gradient stays as it is (block-uniform)
    __shared__ double cv[MAXM];                             // (indexed at run time: LDS, not scratch)
    if ((int)threadIdx.x < m) cv[threadIdx.x] = info[0] == 0 ? 0.0 : (double)(float)v[threadIdx.x];   // v rounded to fp32 as torch.Tensor(x) does downstream
    __syncthreads();
    size_t stride = (size_t)gridDim.x * GB;
    const size_t n4 = VEC ? n / 4 : 0;
    for (size_t c4 = (size_t)blockIdx.x * GB + threadIdx.x; c4 < n4; c4 += stride) {
        const float4 gv = reinterpret_cast<const float4*>(g)[c4];
        double s0 = (double)gv.x, s1 = (double)gv.y, s2 = (double)gv.z, s3 = (double)gv.w;
        for (int i = 0; i < m; ++i) {
            const float4 r = reinterpret_cast<const float4*>(G + (size_t)sel.idx[i] * ld)[c4];
            s0 += cv[i] * (double)r.x; s1 += cv[i] * (double)r.y; s2 += cv[i] * (double)r.z; s3 += cv[i] * (double)r.w;
        }
        reinterpret_cast<float4*>(out)[c4] = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
    }
    for (size_t c = n4 * 4 + (size_t)blockIdx.x * GB + threadIdx.x; c < n; c += stride) {
        double s = (double)g[c];
        for (int i = 0; i < m; ++i) s += cv[i] * (double)G[(size_t)sel.idx[i] * ld + c];
        out[c] = (float)s;
    }
}

template <int M>
int gram_launch(const float* G, size_t ld, const RowSel& sel, size_t n, double* partial, int blocks, bool vec, hipStream_t s) {
    hipLaunchKernelGGL((gram_partial_kernel<M>), dim3(blocks), dim3(GB), 0, s, G, ld, sel, n, vec ? 1 : 0, partial);
    return 0;
}

constexpr int GRAM_BLOCKS = 2048;

}  // namespace

extern "C" {

int clhip_axpy(float* y, const float* x, size_t n, float alpha, int assign, void* stream) {
    if (!y || !x) return CLHIP_EINVAL;
    if (n == 0) return 0;
    if (aligned16(y) && aligned16(x))
        hipLaunchKernelGGL(axpy_kernel<true>, dim3(ew_grid(n / 4 + 1, GB)), dim3(GB), 0, as_stream(stream), y, x, n, alpha, assign);
    else
        hipLaunchKernelGGL(axpy_kernel<false>, dim3(ew_grid(n, GB)), dim3(GB), 0, as_stream(stream), y, x, n, alpha, assign);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

size_t clhip_gem_gram_ws(int m) {
    if (m < 1 || m > MAXM) return 0;
    return (size_t)GRAM_BLOCKS * (m * (m + 1) / 2) * sizeof(double);
}

// out_f64[m*m] (device) = Gram matrix of rows row_idx[0..m) of G (each row n floats, stride ld floats)
int clhip_gem_gram(const float* G, size_t ld, const int* row_idx_host, int m, size_t n, double* out_f64, void* ws,
                   size_t ws_bytes, void* stream) {
    if (!G || !row_idx_host || !out_f64 || !ws || m < 1 || m > MAXM || n == 0 || ws_bytes < clhip_gem_gram_ws(m))
        return CLHIP_EINVAL;
    RowSel sel{};
    for (int i = 0; i < m; ++i) sel.idx[i] = row_idx_host[i];
    hipStream_t s = as_stream(stream);
    const bool vec = aligned16(G) && (ld % 4 == 0);
    int blocks = ew_grid(vec ? n / 4 + 1 : n, GB / gram_split(m));
    if (blocks > GRAM_BLOCKS) blocks = GRAM_BLOCKS;
    double* partial = static_cast<double*>(ws);
    switch (m) {
#define CASE(M_) case M_: gram_launch<M_>(G, ld, sel, n, partial, blocks, vec, s); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16)
#undef CASE
    }
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_final_kernel, dim3(m * (m + 1) / 2), dim3(256), 0, s, partial, blocks, m, out_f64);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// out[c] = g[c] + sum_i v[i] * G[row_idx[i]][c]      (gem.py:79 + overwrite_grad :38-55 when out = grad arena)
int clhip_gem_project(const float* G, size_t ld, const int* row_idx_host, const float* v_host, int m, const float* g,
                      float* out, size_t n, void* stream) {
    if (!G || !row_idx_host || !v_host || !g || !out || m < 1 || m > MAXM || n == 0) return CLHIP_EINVAL;
    RowSel sel{}; Coefs cf{};
    for (int i = 0; i < m; ++i) { sel.idx[i] = row_idx_host[i]; cf.v[i] = v_host[i]; }
    if (aligned16(G) && (ld % 4 == 0) && aligned16(g) && aligned16(out))
        hipLaunchKernelGGL(project_kernel<true>, dim3(ew_grid(n / 4 + 1, GB)), dim3(GB), 0, as_stream(stream), G, ld, sel, cf, m, g, out, n);
    else
        hipLaunchKernelGGL(project_kernel<false>, dim3(ew_grid(n, GB)), dim3(GB), 0, as_stream(stream), G, ld, sel, cf, m, g, out, n);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// v = argmin of GEM's QP, computed on the device from clhip_gem_gram's output (m = memory rows + 1; the current gradient
// is the LAST row).  v_out_f64[m - 1], info_i32[2] = {violated constraints, status}.  Nothing is copied to the host.
int clhip_gem_qp(const double* gram_f64, int m, double margin, double eps, double* v_out_f64, int* info_i32, void* stream) {
    if (!gram_f64 || !v_out_f64 || !info_i32 || m < 2 || m - 1 > QP_MAX || m > MAXM) return CLHIP_EINVAL;
    hipLaunchKernelGGL(gem_qp_kernel, dim3(1), dim3(64), 0, as_stream(stream), gram_f64, m, margin, eps, v_out_f64, info_i32);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// clhip_gem_project with the coefficients (and the "anything violated?" flag) read from device memory: out = g when
// info[0] == 0, else g + sum_i v[i] * G[row_idx[i]].
int clhip_gem_project_dev(const float* G, size_t ld, const int* row_idx_host, const double* v_dev_f64, const int* info_dev, int m,
                          const float* g, float* out, size_t n, void* stream) {
    if (!G || !row_idx_host || !v_dev_f64 || !info_dev || !g || !out || m < 1 || m > MAXM || n == 0) return CLHIP_EINVAL;
    RowSel sel{};
    for (int i = 0; i < m; ++i) sel.idx[i] = row_idx_host[i];
    if (aligned16(G) && (ld % 4 == 0) && aligned16(g) && aligned16(out))
        hipLaunchKernelGGL(project_dev_kernel<true>, dim3(ew_grid(n / 4 + 1, GB)), dim3(GB), 0, as_stream(stream), G, ld, sel, v_dev_f64,
                           info_dev, m, g, out, n);
    else
        hipLaunchKernelGGL(project_dev_kernel<false>, dim3(ew_grid(n, GB)), dim3(GB), 0, as_stream(stream), G, ld, sel, v_dev_f64,
                           info_dev, m, g, out, n);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
