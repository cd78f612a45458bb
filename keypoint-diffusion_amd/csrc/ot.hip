// Exact optimal transport between two small uniform point clouds on the HOST (no kernel here): the plan of
//     min <P, C>   s.t.  P 1 = a,  P^T 1 = b,  P >= 0
// for an [n, m] cost matrix -- what the reference obtains from POT's network simplex (`ot.emd`, losses/rec_encoder_loss.py:11-18)
// once per complex and training batch: n = 20 / 40 keypoints against m ~ 300 receptor atoms (or interface points).  POT is not in
// this image; round 2 solved the program with HiGHS through scipy (0.45 s per complex: 29 s of host time per B = 64 training step
// once the keypoint models could train).  This is a successive-shortest-path min-cost flow on the dense bipartite graph with
// node potentials (reduced costs stay non-negative, every search is one array Dijkstra over the sources), in double precision,
// a few ms per 40 x 300 problem, the complexes of a batch spread over host threads.  The optimal VALUE is unique, so the loss equals
// the reference's; as there, the plan is a constant and the gradient flows through the cost matrix only.
#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>
#include <vector>

#include "common.h"

namespace kpd {
namespace {

// cost [n * m] row-major, masses a [n], b [m] (sums equal); plan [n * m] out.  Returns false on a numerical dead end.
// The searches run over the SOURCES only (n ~ 40 keypoints against m ~ 300 atoms): a path alternates source -> sink -> source, and a
// saturated sink leads on only to the one or two sources that currently serve it, so expanding a source costs one pass over its row
// and a search at most n of them.
bool solve_transport(const double *cost, int n, int m, const double *a, const double *b, double *plan) {
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> pi_s(n, 0.0), pi_t(m), dist(n), dsink(m), supply(a, a + n), demand(b, b + m);
    std::vector<int> prev_sink(n), prev_src(n);
    std::vector<char> done(n);
    // servers of a sink (sources with flow into it): a flat [m][n] table, kept in arrival order
    std::vector<int> srv((size_t)m * n), nsrv(m, 0);
    std::vector<double> scratch(m);
    std::fill(plan, plan + (size_t)n * m, 0.0);
    // feasible potentials: pi_t[j] = min_i c_ij keeps every forward reduced cost c_ij + pi_s[i] - pi_t[j] >= 0
    double total = 0.0, placed = 0.0;
    for (int i = 0; i < n; ++i) total += a[i];
    const double tiny = 1e-14 * std::max(total, 1e-300);
    // greedy start on the arcs that are tight under these potentials (every sink's cheapest source): flow only ever sits on tight
    // arcs, so this is a valid partial solution
    for (int j = 0; j < m; ++j) {
        int best = 0;
        for (int i = 1; i < n; ++i)
            if (cost[(size_t)i * m + j] < cost[(size_t)best * m + j]) best = i;
        pi_t[j] = cost[(size_t)best * m + j];
        const double push = std::min(supply[best], demand[j]);
        if (push > 0.0) {
            plan[(size_t)best * m + j] = push;
            srv[(size_t)j * n + nsrv[j]++] = best;
            supply[best] -= push;
            demand[j] -= push;
            placed += push;
        }
    }
    double remaining = total - placed;
    for (long iter = 0; remaining > tiny; ++iter) {
        if (iter > 64L * (n + m) * (n + m)) return false;
        for (int i = 0; i < n; ++i) { dist[i] = supply[i] > tiny ? 0.0 : INF; done[i] = 0; prev_sink[i] = -1; prev_src[i] = -1; }
        for (int j = 0; j < m; ++j) dsink[j] = INF;
        double best = INF;
        int tj = -1, tsrc = -1;
        for (;;) {
            int u = -1;
            double du = INF;
            for (int i = 0; i < n; ++i)
                if (!done[i] && dist[i] < du) { du = dist[i]; u = i; }
            if (u < 0 || du >= best) break;
            done[u] = 1;
            const double *row = cost + (size_t)u * m;
            // pass 1 (straight-line, vectorisable): tentative distances of all sinks through u
            const double base = du + pi_s[u];
            double *dj_all = scratch.data();
            for (int j = 0; j < m; ++j) {
                const double dj = std::max(base + row[j] - pi_t[j], du);
                dj_all[j] = dj;
                dsink[j] = std::min(dsink[j], dj);
            }
            // pass 2: the sinks that can still improve the best target
            for (int j = 0; j < m; ++j) {
                const double dj = dj_all[j];
                if (dj >= best) continue;
                if (demand[j] > tiny) { best = dj; tj = j; tsrc = u; continue; }
                const int *sj = srv.data() + (size_t)j * n;
                for (int q = 0, nq = nsrv[j]; q < nq; ++q) {      // withdraw flow i2 -> j and carry on from i2
                    const int i2 = sj[q];
                    if (done[i2]) continue;
                    const double nd = dj + std::max(-cost[(size_t)i2 * m + j] + pi_t[j] - pi_s[i2], 0.0);
                    if (nd < dist[i2]) { dist[i2] = nd; prev_sink[i2] = j; prev_src[i2] = u; }
                }
            }
        }
        if (tj < 0) {                // nothing left but rounding residue of the masses (sums of 1 / n vs 1 / m): done
            if (remaining <= 1e-9 * total) break;
            return false;
        }
        for (int i = 0; i < n; ++i) pi_s[i] += std::min(dist[i], best);          // reduced costs stay >= 0, path arcs become tight
        for (int j = 0; j < m; ++j) pi_t[j] += std::min(dsink[j], best);
        // bottleneck along the path tsrc -> tj, back through (prev_sink, prev_src) to a source with supply
        double push = demand[tj];
        int v = tsrc;
        while (prev_sink[v] >= 0) {
            push = std::min(push, plan[(size_t)v * m + prev_sink[v]]);
            v = prev_src[v];
        }
        push = std::min(push, supply[v]);
        if (!(push > 0.0)) return false;
        const int root = v;
        auto add = [&](int i, int j, double f) {
            double &x = plan[(size_t)i * m + j];
            const bool was = x > 0.0;
            x += f;
            if (x <= 0.0) {
                x = 0.0;
                if (was) {                                 // drop i from j's servers, order of the rest kept
                    int *sj = srv.data() + (size_t)j * n;
                    int q = 0;
                    while (sj[q] != i) ++q;
                    for (--nsrv[j]; q < nsrv[j]; ++q) sj[q] = sj[q + 1];
                }
            } else if (!was) srv[(size_t)j * n + nsrv[j]++] = i;
        };
        add(tsrc, tj, push);
        v = tsrc;
        while (prev_sink[v] >= 0) {
            add(v, prev_sink[v], -push);
            add(prev_src[v], prev_sink[v], push);
            v = prev_src[v];
        }
        supply[root] -= push;
        demand[tj] -= push;
        remaining -= push;
    }
    return true;
}

}  // namespace
}  // namespace kpd

// n_problems independent problems: problem p has cost matrix cost + offsets[p] (row-major [n[p], m[p]], doubles) and uniform
// masses 1 / n[p], 1 / m[p]; its plan is written to plan + offsets[p].  Host pointers; spread over up to n_threads threads.
extern "C" kpd_status kpd_ot_emd_uniform(int32_t n_problems, const int32_t *n, const int32_t *m, const int64_t *offsets, const double *cost,
                                         double *plan, int32_t n_threads) {
    KPD_REQUIRE(n_problems >= 0 && (n_problems == 0 || (n && m && offsets && cost && plan)), KPD_ERR_INVALID, "null argument");
    for (int p = 0; p < n_problems; ++p)
        KPD_REQUIRE(n[p] >= 1 && m[p] >= 1, KPD_ERR_INVALID, "optimal transport needs at least one point on either side (problem %d: %d x %d)", p, n[p], m[p]);
    // non-finite costs would still give a plan that satisfies the marginals (comparisons with NaN / inf just fall one way): refuse them
    for (int p = 0; p < n_problems; ++p) {
        const double *c = cost + offsets[p];
        const long long cnt = (long long)n[p] * m[p];
        for (long long i = 0; i < cnt; ++i)
            KPD_REQUIRE(std::isfinite(c[i]), KPD_ERR_INVALID, "optimal-transport problem %d (%d x %d): cost[%lld, %lld] is not finite", p, n[p], m[p],
                        i / m[p], i % m[p]);
    }
    std::vector<char> ok(std::max(n_problems, 1), 1);
    // strided shares [first, first + step, ...): a share that could not get its thread is solved by the caller below
    auto work = [&](int first, int step) {
        for (int p = first; p < n_problems; p += step) {
            std::vector<double> a(n[p], 1.0 / n[p]), b(m[p], 1.0 / m[p]);
            ok[p] = kpd::solve_transport(cost + offsets[p], n[p], m[p], a.data(), b.data(), plan + offsets[p]) ? 1 : 0;
        }
    };
    const int T = std::max(1, std::min<int>(std::min(n_threads, n_problems), 64));
    if (T <= 1) work(0, 1);
    else {
        std::vector<std::thread> th;
        th.reserve(T);
        int started = 0;
        try {                                   // std::thread can throw (resource limits): nothing may cross the extern "C" boundary
            for (; started < T; ++started) th.emplace_back(work, started, T);
        } catch (...) {
        }
        for (int t = started; t < T; ++t) work(t, T);
        for (auto &x : th) x.join();
    }
    for (int p = 0; p < n_problems; ++p)
        KPD_REQUIRE(ok[p], KPD_ERR_INVALID, "optimal-transport problem %d (%d x %d) did not converge (non-finite costs?)", p, n[p], m[p]);
    return KPD_OK;
}
