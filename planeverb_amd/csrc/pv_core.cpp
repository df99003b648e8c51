// pv_core.cpp -- see pv_core.h.  Host-side float32 index arithmetic; no fast-math.
#include "pv_core.h"

#include <cstdio>
#include <cstring>

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iomanip>
#include <limits>
#include <sstream>

namespace pva {

static void fillDerived(GridSpec& g) {
    // PvTypes.h:101 : IR seconds = sqrt2 * 12.5 / c + 0.25, a float constant expression
    const float irSeconds = kSqrt2 * 12.5f / kC + 0.25f;
    g.T = (int)(unsigned)((float)g.fs * irSeconds);            // Grid.cpp:55
    g.courant = kC * g.dt / g.dx;                              // FDTD.cpp:90
    g.nDir = (int)(kDryDirectionLen * (float)g.fs);            // Analyzer.cpp:171
    g.nDry = (int)(kDryGainLen * (float)g.fs);                 // Analyzer.cpp:170
    g.nWet = (int)(kWetGainLen * (float)g.fs);                 // Analyzer.cpp:237
    g.nCut = (int)(kSchroederOffset * (float)g.fs);            // Analyzer.cpp:286
    g.nFree = g.nDry + (int)((1.f / kC) * (float)(int)g.fs);   // FreeGrid.cpp:99
    g.NX = g.gx + 1;
    g.NY = g.gy + 1;
}

static void fillResolution(GridSpec& g, int res) {
    g.res = res;
    const float minWavelength = kC / (float)res;  // Grid.cpp:392
    g.dx = minWavelength / kPointsPerWavelength;  // :393
    g.dt = g.dx / (kC * 1.5f);                    // :394
    g.fs = (unsigned)(1.0f / g.dt);               // :395
}

GridSpec makeGridSpec(float sizeX, float sizeY, int res) {
    GridSpec g;
    fillResolution(g, res);
    g.sizeX = sizeX;
    g.sizeY = sizeY;
    g.gsx = (1.f / g.dx) * sizeX;  // Grid.cpp:48
    g.gsy = (1.f / g.dx) * sizeY;  // Grid.cpp:49
    g.gx = (int)g.gsx;
    g.gy = (int)g.gsy;
    fillDerived(g);
    return g;
}

GridSpec makeGridSpecCells(int gx, int gy, int res) {
    GridSpec g;
    fillResolution(g, res);
    g.gx = gx;
    g.gy = gy;
    g.gsx = (float)gx + 0.5f;
    g.gsy = (float)gy + 0.5f;
    g.sizeX = g.gsx * g.dx;
    g.sizeY = g.gsy * g.dx;
    fillDerived(g);
    return g;
}

std::vector<float> gaussianPulse(const GridSpec& g) {
    // Grid.cpp:12-27.  The 0.5 literal is a double, so sigma is evaluated in double and narrowed; pi is
    // acos(-1) narrowed to float BEFORE use.  expf is the host libm's, as in the reference.
    std::vector<float> out((size_t)g.T);
    const float samplingRate = (float)g.fs;
    const float maxFreq = (float)g.res;
    const float pi = (float)std::acos(-1.0);
    const float sigma = (float)(1.0f / (0.5 * pi * maxFreq));
    const float delay = 2 * sigma;
    const float dt = 1.0f / samplingRate;
    for (int i = 0; i < g.T; ++i) {
        const float t = (float)i * dt;
        out[(size_t)i] = std::exp(-(t - delay) * (t - delay) / (sigma * sigma));
    }
    return out;
}

// Bit parity of the pulse rides on the HOST libm's expf (glibc 2.35 in the reference build): a libm whose
// expf rounds differently would shift every field by an ulp and no test on that host would say why.  Checked once per
// process against five samples of the reference's own 275 Hz table (tests/golden/g71_*.npz `pulse`), incl. a denormal.
bool pulseMatchesReferenceLibm() {
    static const struct {
        int i;
        uint32_t bits;
    } kRef[] = {{0, 0x3c960aaeu}, {10, 0x3ebecae4u}, {20, 0x3405f3a0u}, {30, 0x1c4fb297u}, {40, 0x0000002cu}};
    const std::vector<float> p = gaussianPulse(makeGridSpec(25.f, 25.f, 275));
    for (const auto& r : kRef) {
        uint32_t b;
        std::memcpy(&b, &p[(size_t)r.i], 4);
        if (b != r.bits) return false;
    }
    return true;
}

void warnIfPulseDiffers() {
    static const bool ok = [] {
        const bool m = pulseMatchesReferenceLibm();
        if (!m)
            std::fprintf(stderr, "[planeverb_amd] warning: this host's expf does not reproduce the reference's Gaussian pulse "
                                 "table bit for bit (glibc 2.35 expected): fields and outputs may differ from the reference "
                                 "in the last place\n");
        return m;
    }();
    (void)ok;
}

void listenerCell(const GridSpec& g, float lx, float lz, int* cx, int* cy) {
    *cx = (int)((lx + 0.f) / g.dx);
    *cy = (int)((lz + 0.f) / g.dx);
}

void listenerCellRecip(const GridSpec& g, float lx, float lz, int* cx, int* cy) {
    *cx = (int)(lx * (1.f / g.dx));
    *cy = (int)(lz * (1.f / g.dx));
}

bool resultCell(const GridSpec& g, float ex, float ez, int* cx, int* cy) {
    const unsigned px = (unsigned)((ex + 0.f) / g.dx);
    const unsigned py = (unsigned)((ez + 0.f) / g.dx);
    // The reference tests `>` and so admits one row/column past the gx*gy result map; that read
    // is out of bounds there.  Here it is rejected.
    if (px >= (unsigned)g.gx || py >= (unsigned)g.gy) return false;
    *cx = (int)px;
    *cy = (int)py;
    return true;
}

void MaterialPlane::init(const GridSpec& g) {
    g_ = g;
    const size_t n = (size_t)g.NX * g.NY;
    beta_.assign(n, 1);
    by_.assign(n, 1);
    R_.assign(n, 0.f);
    for (int x = 0; x < g.NX; ++x)
        for (int y = 0; y < g.NY; ++y) {
            if (x == g.gx || y == g.gy) beta_[(size_t)x * g.NY + y] = by_[(size_t)x * g.NY + y] = 0;  // Grid.cpp:93-97
            else if (y == 0) by_[(size_t)x * g.NY + y] = 0;                                            // Grid.cpp:98-102
        }
    markAllDirty();
}

void MaterialPlane::bounds(const Box& b, int* sx, int* sy, int* ex, int* ey) const {
    // Grid.cpp:139-142 : multiply by the reciprocal of dx, truncate toward zero
    const float inv = 1.f / g_.dx;
    *sy = (int)((b.y - b.h / 2.f + 0.f) * inv);
    *sx = (int)((b.x - b.w / 2.f + 0.f) * inv);
    *ey = (int)((b.y + b.h / 2.f + 0.f) * inv);
    *ex = (int)((b.x + b.w / 2.f + 0.f) * inv);
}

void MaterialPlane::add(const Box& b) {
    int sx, sy, ex, ey;
    bounds(b, &sx, &sy, &ex, &ey);
    for (int y = sy; y < ey; ++y) {
        if (!(y >= 0 && (float)y <= g_.gsy)) continue;  // Grid.cpp:231
        for (int x = sx; x < ex; ++x) {
            if (!(x >= 0 && (float)x <= g_.gsx)) continue;  // Grid.cpp:235
            const size_t i = (size_t)x * g_.NY + y;
            R_[i] = b.R;
            beta_[i] = 0;
            by_[i] = 0;  // Grid.cpp:241-242
            dirtyLo_ = std::min(dirtyLo_, x);
            dirtyHi_ = std::max(dirtyHi_, x + 1);
        }
    }
}

void MaterialPlane::remove(const Box& b) {
    int sx, sy, ex, ey;
    bounds(b, &sx, &sy, &ex, &ey);
    for (int y = sy; y < ey; ++y) {
        if (!(y >= 0 && (float)y <= g_.gsy)) continue;
        for (int x = sx; x < ex; ++x) {
            if (!(x >= 0 && (float)x <= g_.gsx)) continue;
            const size_t i = (size_t)x * g_.NY + y;
            R_[i] = 0.f;
            // Grid.cpp:276 tests (y == gx || x == gy); identical to the ghost test on square grids, which
            // are the only self-consistent ones.  The ghost row/column is always beta = 0 here.
            beta_[i] = (x == g_.gx || y == g_.gy) ? 0 : 1;
            // Grid.cpp:281-290: by comes back as 0 on the x == 0 ROW ("j == 0", j being x), not on the y == 0 column
            // the constructor cleared
            by_[i] = (x == g_.gx || y == g_.gy || x == 0) ? 0 : 1;
            dirtyLo_ = std::min(dirtyLo_, x);
            dirtyHi_ = std::max(dirtyHi_, x + 1);
        }
    }
}

void MaterialPlane::clearDirty() {
    dirtyLo_ = g_.NX;
    dirtyHi_ = 0;
}

void MaterialPlane::markAllDirty() {
    dirtyLo_ = 0;
    dirtyHi_ = g_.NX;
}

bool loadPv(const std::string& path, std::vector<Box>* out, std::string* err) {
    std::ifstream f(path);
    if (!f.is_open()) {
        if (err) *err = "cannot open scene file: " + path;
        return false;
    }
    size_t n = 0;
    f >> n;
    out->clear();
    for (size_t i = 0; i < n; ++i) {
        long long id;  // read and discarded, Editor.cpp:271,278
        Box b;
        f >> id >> b.x >> b.y >> b.w >> b.h >> b.R;
        if (!f) {
            if (err) *err = "truncated scene file: " + path;
            return false;
        }
        out->push_back(b);
    }
    return true;
}

bool savePv(const std::string& path, const std::vector<std::pair<int, Box>>& boxes, std::string* err) {
    std::ofstream f(path);
    if (!f.is_open()) {
        if (err) *err = "cannot write scene file: " + path;
        return false;
    }
    // max_digits10: every float survives the text round trip, so a saved scene rasterises to the same cells (the
    // Editor writes 6 significant digits, Editor.cpp:229-241; its operator>> loader reads either form)
    f << std::setprecision(std::numeric_limits<float>::max_digits10);
    f << boxes.size() << std::endl;  // Editor.cpp:229-230
    for (const auto& p : boxes) {
        const Box& b = p.second;
        f << p.first << " " << b.x << " " << b.y << " " << b.w << " " << b.h << " " << b.R << std::endl;
    }
    return true;
}

void reverbBusGains(float rt60, float wet, float* a, float* b, float* c) {
    // three reverb buses with decay times 0.5 / 1.0 / 3.0 s (PvDSPTypes.h:13-15); g(T) = 10^(-3*0.1/T)
    const float t1 = 0.5f, t2 = 1.0f, t3 = 3.0f, tstar = 0.1f;
    auto g = [&](float T) { return std::pow(10.f, -3.f * tstar / T); };
    if (rt60 > t2) *a = 0.f;
    else if (rt60 < t1) *a = 1.f;
    else *a = wet * (g(t2) - g(rt60)) / (g(t2) - g(t1));

    if (rt60 < t1) *b = 0.f;
    else if (rt60 > t2) *b = wet * (g(t3) - g(rt60)) / (g(t3) - g(t2));
    else *b = wet - wet * (g(t2) - g(rt60)) / (g(t2) - g(t1));

    if (rt60 > t3) *c = 1.f;
    else if (rt60 < t2) *c = 0.f;
    else *c = wet - wet * (g(t3) - g(rt60)) / (g(t3) - g(t2));
}

std::vector<SegRect> planSegments(const uint8_t* air, int ntx, int nty, int rxi, int wmax, int target) {
    struct Rect {
        int ti0, nt, tj0, w;
    };
    std::vector<Rect> rects;
    std::vector<SegRect> out;
    if (ntx <= 0 || nty <= 0 || rxi <= 0 || wmax <= 0) return out;
    wmax = std::min(wmax, 7);  // (the chunk key below has 3 bits for the width)
    std::vector<int> open((size_t)nty * 8, -1), next((size_t)nty * 8, -1);
    for (int ti = 0; ti < ntx; ++ti) {
        std::fill(next.begin(), next.end(), -1);
        const uint8_t* row = air + (size_t)ti * nty;
        for (int tj = 0; tj < nty;) {
            if (!row[tj]) {
                ++tj;
                continue;
            }
            int e = tj;
            while (e < nty && row[e]) ++e;
            for (int c = tj; c < e; c += wmax) {
                const int w = std::min(wmax, e - c);
                const size_t key = (size_t)c * 8 + (size_t)w;
                int r = open[key];
                if (r >= 0 && rects[(size_t)r].ti0 + rects[(size_t)r].nt == ti) {
                    ++rects[(size_t)r].nt;
                } else {
                    r = (int)rects.size();
                    rects.push_back({ti, 1, c, w});
                }
                next[key] = r;
            }
            tj = e;
        }
        open.swap(next);
    }
    long long totalRows = 0;
    for (const Rect& r : rects) totalRows += (long long)r.nt * rxi;
    const long long want = (totalRows + std::max(target, 1) - 1) / std::max(target, 1);
    const int Xt = (int)std::min<long long>(7LL * rxi, std::max<long long>(rxi / 2 + 1, want));
    for (const Rect& r : rects) {
        const int rows = r.nt * rxi;
        const int m = (rows + Xt - 1) / Xt;
        for (int k = 0; k < m; ++k) {
            const int r0 = (int)((long long)rows * k / m), r1 = (int)((long long)rows * (k + 1) / m);
            out.push_back(SegRect{r.ti0 * rxi + r0, r1 - r0, r.tj0, r.w});
        }
    }
    std::sort(out.begin(), out.end(), [](const SegRect& x, const SegRect& y) {
        return x.row0 != y.row0 ? x.row0 < y.row0 : x.tj0 < y.tj0;
    });
    return out;
}

}  // namespace pva
