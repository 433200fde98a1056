// C-ABI entry points of the sparse / inducing-state CVI local kernels (mfgm_sparse.h).
#include "mfgm_internal.h"
#include "mfgm_sparse.h"
#include "mfgm_sweeps.h"

using namespace mfgm;

namespace {
SparseArgs sparse_args(const mfgm_sparse_data* s) {
    SparseArgs a;
    a.M = s->M; a.d = s->d; a.N = s->N; a.seg = s->seg;
    a.m_lo = (s->m_hi > 0) ? s->m_lo : 0;
    a.m_hi = (s->m_hi > 0) ? s->m_hi : s->M + 1; a.w = s->w; a.c = s->c; a.prior_mean = s->prior_mean; a.prior_cov = s->prior_cov;
    return a;
}
bool sparse_ok(const mfgm_sparse_data* s) {
    if (!(s && s->M >= 1 && s->d >= 1 && s->d <= 32 && s->N >= 0 && s->seg && (s->N == 0 || (s->w && s->c)))) return false;
    return s->m_hi <= 0 || (s->m_lo >= 0 && s->m_lo < s->m_hi && s->m_hi <= s->M + 1);
}
}  // namespace

extern "C" {

int mfgm_sparse_theta(int T, int d, const double* nat1, const double* nat2, const double* plin, const double* pdiag, const double* psub,
                      double* lin, double* diag, double* sub, void* stream) {
    if (T < 1 || d < 1 || d > 32 || !nat1 || !nat2 || !pdiag || !psub || !lin || !diag || !sub) return 1;
    const size_t total = (size_t)T * d * d;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(k_sparse_theta, dim3(blocks), dim3(256), 0, (hipStream_t)stream, T, d, nat1, nat2, plin, pdiag, psub, lin, diag, sub);
    MFGM_CHECK_LAUNCH();
    return 0;
}

namespace {
int launch_predict(const mfgm_sparse_data* data, const double* mu, const double* Sig, const double* Sub, double* fmu, double* fvar,
                   const SparseKl& kl, hipStream_t st) {
    const int d2 = 2 * data->d;
    const SparseArgs sa = sparse_args(data);
    static const bool vec = [] { const char* e = getenv("MFGM_SPARSE_PREDICT_V"); return !(e && atoi(e) == 0); }();
    if (vec && data->d % 2 == 0 && data->d <= 32) {
        // even d: the coalesced form (rows of a block and the blocks themselves are 16-byte aligned)
        const int np = data->d * data->d / 2;
        if (np <= 128) hipLaunchKernelGGL((k_sparse_predict_v<2>), dim3(sa.m_hi - sa.m_lo), dim3(64), 0, st, sa, mu, Sig, Sub, fmu, fvar, kl);
        else hipLaunchKernelGGL((k_sparse_predict_v<8>), dim3(sa.m_hi - sa.m_lo), dim3(64), 0, st, sa, mu, Sig, Sub, fmu, fvar, kl);
        MFGM_CHECK_LAUNCH();
        return 0;
    }
#define PREDICT(P_) hipLaunchKernelGGL((k_sparse_predict<P_>), dim3(sa.m_hi - sa.m_lo), dim3(64), 0, st, sa, mu, Sig, Sub, fmu, fvar, kl)
    if (d2 <= 2) PREDICT(2); else if (d2 <= 4) PREDICT(4); else if (d2 <= 8) PREDICT(8); else if (d2 <= 16) PREDICT(16);
    else if (d2 <= 32) PREDICT(32); else PREDICT(64);
#undef PREDICT
    MFGM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

int mfgm_sparse_predict(const mfgm_sparse_data* data, const double* mu, const double* Sig, const double* Sub, double* fmu, double* fvar,
                        void* stream) {
    if (!sparse_ok(data) || !mu || !Sig || !Sub || !fmu || !fvar || !data->prior_mean || !data->prior_cov) return 1;
    if (data->N == 0) return 0;
    SparseKl kl;
    memset(&kl, 0, sizeof(kl));
    return launch_predict(data, mu, Sig, Sub, fmu, fvar, kl, (hipStream_t)stream);
}

int mfgm_sparse_predict_kl(const mfgm_sparse_data* data, const double* mu, const double* Sig, const double* Sub, double* fmu, double* fvar,
                           const mfgm_plan* plan, const double* Pd, const double* Ps, double aD, double aS, const double* mup,
                           double* trace, double* maha, void* ws, void* stream) {
    if (!sparse_ok(data) || !mu || !Sig || !Sub || !fmu || !fvar || !data->prior_mean || !data->prior_cov) return 1;
    if (!plan || !Pd || !Ps || !mup || !trace || !maha || !ws) return 1;
    const Plan& P = plan->p;
    if (!P.wide || P.B != 1 || P.T != data->M || P.d != data->d) return 1;
    SparseKl kl{Pd, Ps, mup, aD, aS, (double*)ws + P.off_part[0]};
    int rc = launch_predict(data, mu, Sig, Sub, fmu, fvar, kl, (hipStream_t)stream);
    if (rc) return rc;
    // the sums run over the owned intervals (all of them unless the chain is shared between processes, whose partial sums the caller adds)
    const SparseArgs sa = sparse_args(data);
    return launch_sum_partials(kl.part + sa.m_lo, sa.m_hi - sa.m_lo, data->M + 1, 1, trace, maha, (double*)ws + P.off_part2, (hipStream_t)stream);
}

int mfgm_cond_predict(int M, int d, int N, const int* idx, const double* P, const double* T, const double* prior_mean,
                      const double* prior_cov, const double* mu, const double* Sig, const double* Sub, double* out_mean, double* out_cov,
                      void* stream) {
    if (M < 1 || d < 1 || d > 32 || N < 0 || !idx || !P || !T || !prior_mean || !prior_cov || !mu || !Sig || !Sub || !out_mean || !out_cov)
        return 1;
    if (N == 0) return 0;
    const int d2 = 2 * d;
    const size_t shmem = sizeof(double) * ((size_t)d2 * d2 + d2 + 2 * (size_t)d * d2);
    hipLaunchKernelGGL(k_cond_predict, dim3(N), dim3(64), shmem, (hipStream_t)stream, M, d, N, idx, P, T, prior_mean, prior_cov, mu, Sig, Sub,
                       out_mean, out_cov);
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_sparse_site_update_q(const mfgm_sparse_data* data, const double* g1, const double* g2, double lr, double* nat1, double* nat2q,
                              void* stream) {
    if (!sparse_ok(data) || !nat1 || !nat2q || (data->N > 0 && (!g1 || !g2))) return 1;
    const int d = data->d, QS = d * (d + 1) + d * d;
    const size_t shmem = sizeof(double) * kSitesQChunk * (2 * d + 2);
    const SparseArgs sa = sparse_args(data);
    const int per_wg = kSitesQG * kSitesQRounds;
    const dim3 grid((sa.m_hi - sa.m_lo + per_wg - 1) / per_wg);
    const int ne = (QS + 255) / 256;
#define SITESQ(NE_) hipLaunchKernelGGL((k_sparse_sites_q<NE_>), grid, dim3(256), shmem, (hipStream_t)stream, sa, g1, g2, lr, nat1, nat2q)
    if (ne <= 1) SITESQ(1); else if (ne <= 2) SITESQ(2); else if (ne <= 3) SITESQ(3); else if (ne <= 4) SITESQ(4);
    else if (ne <= 8) SITESQ(8); else return 1;
#undef SITESQ
    MFGM_CHECK_LAUNCH();
    return 0;
}

int mfgm_sparse_site_update(const mfgm_sparse_data* data, const double* g1, const double* g2, double lr, double* nat1, double* nat2,
                            void* stream) {
    if (!sparse_ok(data) || !nat1 || !nat2 || (data->N > 0 && (!g1 || !g2))) return 1;
    const int npair = 2 * data->d * data->d;            // entry pairs of a [2d, 2d] block
    const size_t shmem = sizeof(double) * kSitesChunk * (2 * data->d + 2);
    const SparseArgs sa = sparse_args(data);
#define SITES(SPI_)                                                                                                                    \
    hipLaunchKernelGGL((k_sparse_sites<SPI_>), dim3((sa.m_hi - sa.m_lo + 8 / SPI_ - 1) / (8 / SPI_)), dim3(256), shmem, (hipStream_t)stream,  \
                       sa, g1, g2, lr, nat1, nat2)
    if (npair <= 256) SITES(1); else if (npair <= 512) SITES(2); else if (npair <= 1024) SITES(4); else SITES(8);
#undef SITES
    MFGM_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
