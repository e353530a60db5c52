// (eps, L) adapter: host C++ restatement of tensorBNN/paramAdapter.py:11-292
// (GP-UCB over an eps x L grid; reward = squared jump distance / sqrt(L)).
// The reference runs this on the host too (TF eager + one jitted grid loop),
// once per epoch; it needs no GPU.  float32 arithmetic like the reference
// (paramAdapter.py:60); the matrix inverse is done in double and rounded.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <random>
#include <string>
#include <thread>
#include <cstdlib>
#include <vector>

#include "../../include/tbnn.h"

namespace {

struct Adapter {
    float currentE, currentL;
    float el, eu, Ll, Lu;
    int eNumber;
    std::vector<float> eGrid, lGrid;
    float delta, a;
    float sig0 = 6.25f, sig1 = 6.25f;          // diag(1/(kappa*2)^2), kappa = 0.2   (:72-74)
    std::vector<std::pair<float, float>> previousGamma;
    std::vector<float> allSD, allData, currentData;
    std::vector<float> K;                       // row-major n x n
    int nK = 0;
    double k;                                   // burnin / averagingSteps (may be fractional)
    int m;
    float maxR = 1e-8f;
    long i = -2;                                // :85
    std::vector<float> prev, cur;
    bool havePrev = false, haveCur = false;
    int strikes = 0, maxStrikes = 50;           // :92 (the `strikes` ctor argument is ignored)
    int randomSteps;
    std::mt19937_64 rng;
    // per-update scratch
    double p = 1.0;
    float s = 0.f, rootbeta = 0.f;
    std::vector<float> inverse, inverseR;

    void linspace() {                           // numpy/tf linspace, :68
        eGrid.resize(eNumber);
        const double st = eNumber > 1 ? ((double)eu - (double)el) / (eNumber - 1) : 0.0;
        for (int j = 0; j < eNumber; ++j) eGrid[j] = (float)((double)el + j * st);
        if (eNumber > 1) eGrid[eNumber - 1] = eu;
    }
    void reset() {                              // :143-156
        previousGamma.clear(); allSD.clear(); K.clear(); nK = 0;
        currentData.clear(); allData.clear();
        maxR = 1e-8f; i = -2; havePrev = haveCur = false; strikes = 0;
    }
    float scaleE(float e) const { return -1.f + 2.f * (e - el) / (eu - el); }
    float scaleL(float L) const { return -1.f + 2.f * (L - Ll) / (Lu - Ll); }
    // :95-111 -- exp(-1/2 g1^T Sigma g2): a dot-product form (Q8)
    float calck(std::pair<float, float> gi, std::pair<float, float> gj) const {
        const float d = scaleE(gi.first) * sig0 * scaleE(gj.first) + scaleL(gi.second) * sig1 * scaleL(gj.second);
        return std::exp(-0.5f * d);
    }
    // :113-141 -- ucb = mean + variance * p * rootbeta (variance, not its root: Q9)
    float ucb(float e, float L, std::vector<float>& kv) const {
        const int h = (int)previousGamma.size();
        for (int j = 0; j < h; ++j) kv[j] = calck(previousGamma[j], {e, L});
        float mean = 0.f;
        for (int j = 0; j < h; ++j) mean += kv[j] * inverseR[j];
        mean *= s;
        float quad = 0.f;
        for (int r = 0; r < h; ++r) {
            float t = 0.f;
            for (int c = 0; c < h; ++c) t += inverse[r * h + c] * kv[c];
            quad += kv[r] * t;
        }
        const float var = calck({e, L}, {e, L}) - quad;
        return mean + var * (float)p * rootbeta;
    }
    // :158-196 -- e fastest, then L; keeps the first strictly greater ucb; init -1e9.
    // The grid (trainRegression.py: 100 x 991 points x a 50 x 50 quadratic form = 2.5e8 FMAs, 30-80 ms on one core
    // every averagingSteps epochs -- as much as ten configs[1] epochs on the GPU) is scanned by a few host threads:
    // every thread takes a contiguous block of L rows and keeps the first strictly greater value of its block, the
    // blocks are merged in order with the same strict comparison -- the sequential scan's result, bit for bit.
    struct Best { float u, e, L; };
    Best scanRows(size_t l0, size_t l1) const {
        std::vector<float> kv(previousGamma.size());
        Best b{-1e9f, el, Ll};
        for (size_t li = l0; li < l1; ++li)
            for (int ei = 0; ei < eNumber; ++ei) {
                const float u = ucb(eGrid[ei], lGrid[li], kv);
                if (u > b.u) { b.u = u; b.e = eGrid[ei]; b.L = lGrid[li]; }
            }
        return b;
    }
    void gridSearch() {
        const size_t rows = lGrid.size();
        const size_t work = rows * (size_t)eNumber * previousGamma.size() * previousGamma.size();
        unsigned nt = 1;
        if (work > (1u << 22)) {
            nt = std::thread::hardware_concurrency();
            if (const char* e = std::getenv("TBNN_ADAPTER_THREADS")) nt = (unsigned)std::max(1, std::atoi(e));
            else nt = std::min(nt == 0 ? 1u : nt, 16u);             // 8 ranks per node share the host
            nt = (unsigned)std::min<size_t>(nt, rows);
        }
        std::vector<Best> part(nt);
        if (nt <= 1) part[0] = scanRows(0, rows);
        else {
            std::vector<std::thread> th;
            for (unsigned t = 0; t < nt; ++t)
                th.emplace_back([&, t]() { part[t] = scanRows(rows * t / nt, rows * (t + 1) / nt); });
            for (auto& x : th) x.join();
        }
        Best b{-1e9f, el, Ll};
        for (unsigned t = 0; t < nt; ++t) if (part[t].u > b.u) b = part[t];
        currentE = b.e; currentL = b.L;
    }
    // (K + sn2 I)^-1 ; returns false when singular
    bool invert(float sn2, float extra) {
        const int n = nK;
        std::vector<double> A((size_t)n * 2 * n, 0.0);
        for (int r = 0; r < n; ++r) {
            for (int c = 0; c < n; ++c) A[(size_t)r * 2 * n + c] = (double)K[r * n + c] + (r == c ? (double)sn2 + extra : 0.0);
            A[(size_t)r * 2 * n + n + r] = 1.0;
        }
        for (int c = 0; c < n; ++c) {
            int piv = c;
            for (int r = c + 1; r < n; ++r)
                if (std::fabs(A[(size_t)r * 2 * n + c]) > std::fabs(A[(size_t)piv * 2 * n + c])) piv = r;
            if (std::fabs(A[(size_t)piv * 2 * n + c]) < 1e-300) return false;
            if (piv != c) for (int j = 0; j < 2 * n; ++j) std::swap(A[(size_t)piv * 2 * n + j], A[(size_t)c * 2 * n + j]);
            const double d = A[(size_t)c * 2 * n + c];
            for (int j = 0; j < 2 * n; ++j) A[(size_t)c * 2 * n + j] /= d;
            for (int r = 0; r < n; ++r) {
                if (r == c) continue;
                const double f = A[(size_t)r * 2 * n + c];
                if (f == 0.0) continue;
                for (int j = 0; j < 2 * n; ++j) A[(size_t)r * 2 * n + j] -= f * A[(size_t)c * 2 * n + j];
            }
        }
        inverse.assign((size_t)n * n, 0.f);
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) inverse[r * n + c] = (float)A[(size_t)r * 2 * n + n + c];
        return true;
    }

    // paramAdapter.update, :199-292
    void update(const float* state, int P, float inject_u, int inject_e, int inject_l, float* sjd_out) {
        if ((double)i < k - 2 && strikes == maxStrikes) {               // :208-214
            el = el / 2.f; eu = eu / 2.f;
            linspace();
            k = k - (double)i - 2;
            reset();
            strikes = 0;
        }
        prev.swap(cur); havePrev = haveCur;
        cur.assign(state, state + P); haveCur = true;
        float sjd = 0.f;
        if (havePrev) {                                                 // :218-228
            float acc = 0.f;
            for (int j = 0; j < P; ++j) { const float d = cur[j] - prev[j]; acc += d * d; }
            sjd = acc / std::sqrt(currentL);
            currentData.push_back(sjd);
            const long q = (long)std::floor((double)i / m);             // python floor division
            if (sjd < 1e-8f && q > randomSteps) strikes += 1; else strikes = 0;
        }
        if (sjd_out) *sjd_out = sjd;
        const long mod = ((i % m) + m) % m;                             // python modulo
        if (mod == 0 && i > 0) {                                        // :231
            float u = inject_u;
            if (u < 0.f) u = std::uniform_real_distribution<float>(0.f, 1.f)(rng);   // :232
            p = std::pow(std::max((double)i / m - k + 1.0, 1.0), -0.5);  // :233
            if ((double)u < p) {
                const int nd = (int)currentData.size();
                float mean = 0.f;
                for (float v : currentData) mean += v;
                mean /= (float)nd;
                float var = 0.f;
                for (float v : currentData) var += (v - mean) * (v - mean);
                const float sd = std::sqrt(var / (float)nd);            // reduce_std: population
                currentData.clear();
                allData.push_back(mean); allSD.push_back(sd);
                maxR = *std::max_element(allData.begin(), allData.end());
                previousGamma.push_back({currentE, currentL});          // :242
                const int size = (int)previousGamma.size();
                std::vector<float> nK((size_t)size * size, 0.f);        // :244-257 grow K by one row/col
                for (int r = 0; r < this->nK; ++r)
                    for (int c = 0; c < this->nK; ++c) nK[r * size + c] = K[r * this->nK + c];
                for (int j = 0; j < size; ++j) {
                    const float kk = calck(previousGamma[j], previousGamma[size - 1]);
                    nK[(size - 1) * size + j] = kk; nK[j * size + size - 1] = kk;
                }
                K.swap(nK); this->nK = size;
                s = a / maxR;                                           // :258
                float sn = 0.f;
                for (float v : allSD) sn += v;
                sn /= (float)allSD.size();
                if (!invert(sn * sn, 0.0)) invert(sn * sn, 0.1);        // :263-269
                inverseR.assign(size, 0.f);                             // :270
                for (int r = 0; r < size; ++r) {
                    float t = 0.f;
                    for (int c = 0; c < size; ++c) t += inverse[r * size + c] * allData[c];
                    inverseR[r] = t;
                }
                const double rb0 = std::pow((double)i / m + 1.0, 3.0) * M_PI * M_PI;   // :274
                float rb = (float)rb0 / (3.f * delta);                  // :275
                rb = std::log(rb) * 2.f;                                // :276
                rootbeta = std::sqrt(rb);                               // :277
                const long q = (long)std::floor((double)i / m);
                if (q >= randomSteps) gridSearch();                     // :280-281
                else {                                                  // :283-284 random.choice
                    const int ei = inject_e >= 0 ? inject_e % eNumber
                                                 : (int)std::uniform_int_distribution<int>(0, eNumber - 1)(rng);
                    const int nl = (int)lGrid.size();
                    const int li = inject_l >= 0 ? inject_l % nl : (int)std::uniform_int_distribution<int>(0, nl - 1)(rng);
                    currentE = eGrid[ei]; currentL = lGrid[li];
                }
                if (size == 50) {                                       // :285-289 sliding window
                    std::vector<float> K2((size_t)49 * 49);
                    for (int r = 1; r < 50; ++r)
                        for (int c = 1; c < 50; ++c) K2[(r - 1) * 49 + (c - 1)] = K[r * 50 + c];
                    K.swap(K2); this->nK = 49;
                    previousGamma.erase(previousGamma.begin());
                    allData.erase(allData.begin());
                    allSD.erase(allSD.begin());
                }
            }
        }
        i += 1;                                                         // :291
    }
};

thread_local std::string g_aerr;

}  // namespace

struct tbnn_adapter { Adapter a; };

extern "C" int tbnn_adapter_create(float e1, int32_t L1, float el, float eu, int32_t eNumber, int32_t Ll, int32_t Lu,
                                   int32_t lStep, int32_t m, double k, float a, float delta, int32_t randomSteps,
                                   uint64_t seed, tbnn_adapter_handle* out) {
    if (!out) return -1;
    *out = nullptr;
    if (eNumber < 1 || lStep < 1 || Lu < Ll || m < 1 || !(eu > el)) return -1;
    tbnn_adapter* h = new tbnn_adapter();
    Adapter& A = h->a;
    A.currentE = e1; A.currentL = (float)L1;
    A.el = el; A.eu = eu; A.Ll = (float)Ll; A.Lu = (float)Lu;
    A.eNumber = eNumber;
    A.linspace();
    for (int L = Ll; L <= Lu; L += lStep) A.lGrid.push_back((float)L);   // :69
    A.delta = delta; A.a = a; A.k = k; A.m = m; A.randomSteps = randomSteps;
    A.rng.seed(seed);
    *out = h;
    return 0;
}

extern "C" int tbnn_adapter_destroy(tbnn_adapter_handle a) { delete a; return 0; }

extern "C" int tbnn_adapter_update(tbnn_adapter_handle a, const float* state, int32_t P, float inject_u, int32_t inject_e,
                                   int32_t inject_l, float* eps_out, int32_t* L_out, float* sjd_out) {
    if (!a || !state || P < 1) return -1;
    a->a.update(state, P, inject_u, inject_e, inject_l, sjd_out);
    if (eps_out) *eps_out = a->a.currentE;
    if (L_out) *L_out = (int32_t)a->a.currentL;
    return 0;
}
