// Monte-Carlo interventional target of an additive structural equation model.
//
// Replaces the Python loop of compute_interventions (/root/reference/src/utils_functions/graph_functions.py:48-77:
// num_samples x sample_from_model (:8-27) on the mutilated model of intervene_dict (:30-45), then the mean of
// the target column).  One thread walks the nodes of one noise draw in topological order; the node values
// of a draw live in LDS (the parent index is a run-time value, registers cannot be indexed by it).
// The noise matrix is stored column-major on the device so that consecutive threads read consecutive
// doubles.  Sums: thread partial -> wave (DPP shuffles) -> workgroup -> partial[intervention][block], and a
// second kernel adds the block partials in index order, so the result does not depend on scheduling.
//
// Compute-bound on fp64 transcendentals (about n_terms exp/cos/sin per draw); HBM traffic is the noise
// matrix once per group of interventions (it stays in L2/MALL: 100000 x 9 doubles = 7.2 MB).
#include "cbo_internal.h"

namespace cbo {

constexpr int kSemThreads = 256;
constexpr int kSemDrawsPerThread = 4;     // draws of one intervention handled by one thread

__device__ __forceinline__ double sem_fn(int fn, double x)
{
    switch (fn) {
    case CBO_FN_SQUARE: return x * x;
    case CBO_FN_EXP: return exp(x);
    case CBO_FN_COS: return cos(x);
    case CBO_FN_SIN: return sin(x);
    default: return x;
    }
}

__global__ __launch_bounds__(kSemThreads) void sem_mc_kernel(cbo_sem_spec spec, const double *__restrict__ eps_cm,
                                                              int64_t n_draws, int target, unsigned live, int n_iv,
                                                              const int *__restrict__ iv_nodes,
                                                              const double *__restrict__ iv_values, int n_blocks,
                                                              double *__restrict__ partial)
{
    __shared__ double val[CBO_SEM_MAX_NODES][kSemThreads];
    __shared__ double wsum[kSemThreads / 64];
    __shared__ double fixed[CBO_SEM_MAX_NODES];
    __shared__ int is_fixed[CBO_SEM_MAX_NODES];
    const int tid = threadIdx.x;
    const int64_t iv = blockIdx.y;
    if (tid < CBO_SEM_MAX_NODES) is_fixed[tid] = 0;
    __syncthreads();
    if (tid < n_iv) {                                  // later entries win, like dict.update in intervene_dict
        is_fixed[iv_nodes[tid]] = 1;
    }
    __syncthreads();
    if (tid == 0)
        for (int j = 0; j < n_iv; ++j) fixed[iv_nodes[j]] = iv_values[iv * n_iv + j];
    __syncthreads();

    double acc = 0.0;
    for (int rep = 0; rep < kSemDrawsPerThread; ++rep) {
        const int64_t s = ((int64_t)blockIdx.x * kSemDrawsPerThread + rep) * kSemThreads + tid;
        if (s >= n_draws) break;
        for (int k = 0; k <= target; ++k) {
            if (!((live >> k) & 1u)) continue;        // the target does not depend on this node under this do()
            double v;
            if (is_fixed[k]) {
                v = fixed[k];
            } else {
                const int tb = spec.term_begin[k], te = spec.term_begin[k + 1];
                v = 0.0;
                for (int t = tb; t < te; ++t) {
                    const double x = val[spec.term_parent[t]][tid];
                    const double g = spec.term_c[t] * sem_fn(spec.term_fn[t], spec.term_a[t] * x);
                    v = (t == tb) ? g : v + g;        // left to right, the noise last, as the reference writes them
                }
                if (spec.eps_index[k] >= 0) {
                    const double e = eps_cm[(int64_t)spec.eps_index[k] * n_draws + s];
                    v = (te > tb) ? v + e : e;
                }
            }
            val[k][tid] = v;
        }
        acc += val[target][tid];
    }
    // workgroup sum in a fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((tid & 63) == 0) wsum[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = wsum[0];
        for (int w = 1; w < kSemThreads / 64; ++w) t += wsum[w];
        partial[iv * n_blocks + blockIdx.x] = t;
    }
}

__global__ void sem_mean_kernel(const double *__restrict__ partial, int n_blocks, int64_t m, int64_t n_draws,
                                double *__restrict__ mean_out)
{
    const int64_t iv = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (iv >= m) return;
    double t = 0.0;
    for (int b = 0; b < n_blocks; ++b) t += partial[iv * n_blocks + b];
    mean_out[iv] = t / (double)n_draws;
}

int sem_partial_blocks(int64_t n_draws)
{
    const int64_t per_block = (int64_t)kSemThreads * kSemDrawsPerThread;
    return (int)((n_draws + per_block - 1) / per_block);
}

void launch_sem_target(hipStream_t s, const cbo_sem_spec &spec, const double *eps_cm, int64_t n_draws, int target,
                       int64_t m, int n_iv, const int *iv_nodes_host, const int *iv_nodes, const double *iv_values,
                       double *partial, double *mean_out)
{
    const int nb = sem_partial_blocks(n_draws);
    // nodes the target depends on once the intervened ones are cut from their parents (walk the terms backwards)
    unsigned live = 1u << target, cut = 0;
    for (int j = 0; j < n_iv; ++j) cut |= 1u << iv_nodes_host[j];
    for (int k = target; k >= 0; --k)
        if (((live >> k) & 1u) && !((cut >> k) & 1u))
            for (int t = spec.term_begin[k]; t < spec.term_begin[k + 1]; ++t) live |= 1u << spec.term_parent[t];
    // grid.y is limited to 65535: interventions go out in slabs
    for (int64_t i0 = 0; i0 < m; i0 += 65535) {
        const int64_t cnt = (m - i0 < 65535) ? (m - i0) : 65535;
        hipLaunchKernelGGL(sem_mc_kernel, dim3(nb, (unsigned)cnt), dim3(kSemThreads), 0, s, spec, eps_cm, n_draws,
                           target, live, n_iv, iv_nodes, iv_values + i0 * n_iv, nb, partial + i0 * nb);
    }
    hipLaunchKernelGGL(sem_mean_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, partial, nb, m, n_draws,
                       mean_out);
}

}  // namespace cbo
