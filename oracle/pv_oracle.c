/* oracle/pv_oracle.c -- TEST INFRASTRUCTURE ONLY.  See pv_oracle.h.
 *
 * CPU restatement of the reference algorithm for the FDTD + IR-analysis hot path.  Parity is PINNED:
 * tests/test_oracle_vs_ref.py compares every output of this file bit-for-bit with the unmodified reference
 * compiled into oracle/_ref/libpvref.so, and tests/golden/ holds vectors generated from that build.
 * Nothing here is linked into, imported by or executed from the product (planeverb_amd/).
 */
#include "pv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* PvTypes.h:83-101 */
#define PV_C 343.21f
#define PV_AUDIBLE_THRESHOLD_GAIN 0.00000316f
#define PV_DRY_DIRECTION_ANALYSIS_LENGTH 0.005f
#define PV_DRY_GAIN_ANALYSIS_LENGTH 0.01f
#define PV_WET_GAIN_ANALYSIS_LENGTH 0.080f
#define PV_SQRT_2 1.4142136f
#define PV_POINTS_PER_WAVELENGTH 3.5f
#define PV_SCHROEDER_OFFSET_S 0.01f
#define PV_DISTANCE_GAIN_THRESHOLD 0.891251f
#define PV_DELAY_CLOSE_THRESHOLD 5.f

/* Grid.cpp:390-396 */
void pvo_grid_params(int res, float* dx, float* dt, unsigned* fs) {
    float minWavelength = PV_C / (float)res;
    *dx = minWavelength / PV_POINTS_PER_WAVELENGTH;
    *dt = *dx / (PV_C * 1.5f);
    *fs = (unsigned)(1.0f / *dt);
}

/* PvTypes.h:101 (constexpr float expression) and Grid.cpp:55 */
int pvo_response_length(unsigned fs) {
    const float irs = PV_SQRT_2 * 12.5f / PV_C + 0.25f;
    return (int)(unsigned)((float)fs * irs);
}

/* Grid.cpp:12-27.  sigma is evaluated in double (0.5 is a double literal) and narrowed; pi = acos(-1)
 * narrowed to float first. */
void pvo_gaussian_pulse(int res, unsigned fs, float* out, int n) {
    const float samplingRate = (float)fs;
    const float maxFreq = (float)res;
    const float pi = (float)acos(-1.0);
    float sigma = (float)(1.0f / (0.5 * pi * maxFreq));
    const float delay = 2 * sigma;
    const float dt = 1.0f / samplingRate;
    for (int i = 0; i < n; ++i) {
        float t = (float)i * dt;
        out[i] = expf(-(t - delay) * (t - delay) / (sigma * sigma));
    }
}

/* Grid.cpp:30-117 */
PvoGrid* pvo_grid_create(float sizeX, float sizeY, int res, int with_history) {
    PvoGrid* g = (PvoGrid*)calloc(1, sizeof(PvoGrid));
    g->res = res;
    g->sizeX = sizeX;
    g->sizeY = sizeY;
    pvo_grid_params(res, &g->dx, &g->dt, &g->fs);
    g->gridSizeXf = (1.f / g->dx) * sizeX; /* Grid.cpp:48-49 */
    g->gridSizeYf = (1.f / g->dx) * sizeY;
    g->gx = (int)g->gridSizeXf;
    g->gy = (int)g->gridSizeYf;
    g->ncell = (int)((unsigned)(g->gridSizeXf + 1) * (unsigned)(g->gridSizeYf + 1)); /* Grid.cpp:53 */
    g->T = pvo_response_length(g->fs);
    g->b = (short*)malloc(sizeof(short) * g->ncell);
    g->R = (float*)malloc(sizeof(float) * g->ncell);
    g->pulse = (float*)malloc(sizeof(float) * g->T);
    /* Grid.cpp:84-108: R = free space everywhere; b = 0 on the ghost row x==gx / ghost column y==gy */
    const int incY = (int)(g->gridSizeYf + 1);
    for (int i = 0; i < g->ncell; ++i) {
        int row = i / incY, col = i % incY;
        g->R[i] = 0.f;
        g->b[i] = (row == g->gx || col == g->gy) ? 0 : 1;
    }
    pvo_gaussian_pulse(res, g->fs, g->pulse, g->T);
    if (with_history) {
        size_t n = (size_t)g->T * g->ncell;
        g->hist_pr = (float*)calloc(n, sizeof(float));
        g->hist_vx = (float*)calloc(n, sizeof(float));
        g->hist_vy = (float*)calloc(n, sizeof(float));
    }
    return g;
}

void pvo_grid_destroy(PvoGrid* g) {
    if (!g) return;
    free(g->b);
    free(g->R);
    free(g->pulse);
    free(g->hist_pr);
    free(g->hist_vx);
    free(g->hist_vy);
    free(g);
}

/* Grid.cpp:136-144 bounds (gridWorldOffset is zero: "not supported", PvTypes.h:58-59) */
static void aabb_bounds(const PvoGrid* g, const float* a, int* sx, int* sy, int* ex, int* ey) {
    const float inv = 1.f / g->dx;
    *sy = (int)((a[1] - a[3] / 2.f + 0.f) * inv);
    *sx = (int)((a[0] - a[2] / 2.f + 0.f) * inv);
    *ey = (int)((a[1] + a[3] / 2.f + 0.f) * inv);
    *ex = (int)((a[0] + a[2] / 2.f + 0.f) * inv);
}

/* Grid.cpp:229-246 : index = INDEX(x, y, (gx+1, gy+1)) = x*(gx+1) + y  (stride gx+1: SURVEY quirk Q1) */
void pvo_add_aabb(PvoGrid* g, const float* a) {
    int sx, sy, ex, ey;
    aabb_bounds(g, a, &sx, &sy, &ex, &ey);
    const unsigned dimx = (unsigned)(g->gridSizeXf + 1);
    for (int i = sy; i < ey; ++i) {
        if (i >= 0 && (float)i <= g->gridSizeYf) {
            for (int j = sx; j < ex; ++j) {
                if (j >= 0 && (float)j <= g->gridSizeXf) {
                    int index = (int)(j * dimx + i);
                    g->R[index] = a[4];
                    g->b[index] = 0;
                }
            }
        }
    }
}

/* Grid.cpp:249-296 : restores air, re-zeroing ghosts with the reference's swapped test (Grid.cpp:276) */
void pvo_remove_aabb(PvoGrid* g, const float* a) {
    int sx, sy, ex, ey;
    aabb_bounds(g, a, &sx, &sy, &ex, &ey);
    const unsigned dimx = (unsigned)(g->gridSizeXf + 1);
    for (int i = sy; i < ey; ++i) {
        if (i >= 0 && (float)i <= g->gridSizeYf) {
            for (int j = sx; j < ex; ++j) {
                if (j >= 0 && (float)j <= g->gridSizeXf) {
                    int index = (int)(j * dimx + i);
                    g->R[index] = 0.f;
                    g->b[index] = (i == g->gx || j == g->gy) ? 0 : 1;
                }
            }
        }
    }
}

/* FDTD.cpp:97-99 */
void pvo_listener_cell(const PvoGrid* g, float lx, float lz, int* cx, int* cy) {
    *cx = (int)((lx + 0.f) / g->dx);
    *cy = (int)((lz + 0.f) / g->dx);
}

/* FDTD.cpp:87-236.  Reads past the array end in the pressure sweep (FDTD.cpp:134,136)
 * are multiplied by beta = 0 in the reference; here the two arrays carry S+1 zero cells of slack. */
void pvo_fdtd(PvoGrid* g, float lx, float lz, float* fields) {
    const float Courant = PV_C * g->dt / g->dx; /* FDTD.cpp:90 */
    const int gridx = g->gx, gridy = g->gy;
    const int S = gridy + 1;
    const int N = g->ncell;
    int lcx, lcy;
    pvo_listener_cell(g, lx, lz, &lcx, &lcy);
    const int listenerPos = lcx * S + lcy;
    float* pr = (float*)calloc((size_t)N + S + 2, sizeof(float));
    float* vx = (float*)calloc((size_t)N + S + 2, sizeof(float));
    float* vy = (float*)calloc((size_t)N + S + 2, sizeof(float));
    const short* b = g->b;
    const float* R = g->R;

    for (int t = 0; t < g->T; ++t) {
        /* pressure: FDTD.cpp:124-141 */
        for (int i = 0; i < N; ++i) {
            float beta = (float)(int)b[i];
            float divergence = ((vx[i + S] - vx[i]) + (vy[i + 1] - vy[i]));
            pr[i] = beta * (pr[i] - Courant * divergence);
        }
        /* vx: FDTD.cpp:143-170 */
        for (int i = S; i < N; ++i) {
            int in = i - S;
            float beta_n = (float)b[in];
            float Rn = R[in];
            float Yn = (1.f - Rn) / (1.f + Rn);
            float beta = (float)(int)b[i];
            float Rc = R[i];
            float Y = (1.f - Rc) / (1.f + Rc);
            float gradient_x = (pr[i] - pr[in]);
            float airCellUpdate = vx[i] - Courant * gradient_x;
            float Y_boundary = beta * Yn + beta_n * Y;
            float wallCellUpdate = Y_boundary * (pr[in] * beta_n + pr[i] * beta);
            vx[i] = beta * beta_n * airCellUpdate + (beta_n - beta) * wallCellUpdate;
        }
        /* vy: FDTD.cpp:172-199 */
        for (int i = 1; i < N; ++i) {
            int in = i - 1;
            float beta_n = (float)b[in];
            float Rn = R[in];
            float Yn = (1.f - Rn) / (1.f + Rn);
            float beta = (float)(int)b[i];
            float Rc = R[i];
            float Y = (1.f - Rc) / (1.f + Rc);
            float gradient_y = (pr[i] - pr[in]);
            float airCellUpdate = vy[i] - Courant * gradient_y;
            float Y_boundary = beta * Yn + beta_n * Y;
            float wallCellUpdate = Y_boundary * (pr[in] * beta_n + pr[i] * beta);
            vy[i] = beta * beta_n * airCellUpdate + (beta_n - beta) * wallCellUpdate;
        }
        /* absorbing edges: FDTD.cpp:201-223 */
        for (int i = 0; i < gridy; ++i) {
            int index1 = i;
            int index2 = gridx * (gridy + 1) + i;
            vx[index1] = -pr[index1];
            vx[index2] = pr[index2 - gridy - 1];
        }
        for (int i = 0; i < gridx; ++i) {
            int index1 = i * (gridy + 1);
            int index2 = i * (gridy + 1) + gridy;
            vy[index1] = -pr[index1];
            vy[index2] = pr[index2 - 1];
        }
        /* record: FDTD.cpp:226-230 */
        if (g->hist_pr) {
            memcpy(g->hist_pr + (size_t)t * N, pr, sizeof(float) * N);
            memcpy(g->hist_vx + (size_t)t * N, vx, sizeof(float) * N);
            memcpy(g->hist_vy + (size_t)t * N, vy, sizeof(float) * N);
        }
        /* pulse: FDTD.cpp:234 */
        pr[listenerPos] += g->pulse[t];
    }
    if (fields) {
        memcpy(fields, pr, sizeof(float) * N);
        memcpy(fields + N, vx, sizeof(float) * N);
        memcpy(fields + 2 * (size_t)N, vy, sizeof(float) * N);
    }
    free(pr);
    free(vx);
    free(vy);
}

/* FreeGrid.cpp:41-59 */
float pvo_efree_per_r(float efree, float dx, int lX, int lY, int eX, int eY) {
    float lx = (float)lX * dx, ly = (float)lY * dx;
    float ex = (float)eX * dx, ey = (float)eY * dx;
    float r = sqrtf((ex - lx) * (ex - lx) + (ey - ly) * (ey - ly));
    if (r == 0.f) return efree;
    return efree / r;
}

/* FreeGrid.cpp:6-34,71-110 */
float pvo_free_energy(float sizeX, float sizeY, int res) {
    PvoGrid* g = pvo_grid_create(sizeX, sizeY, res, 1);
    int gridx = g->gx, gridy = g->gy;
    int listenerX = gridx / 2, listenerY = gridy / 2;
    int emitterX = listenerX + (int)(1.f / g->dx);
    int emitterY = listenerY;
    pvo_fdtd(g, (float)listenerX * g->dx, (float)listenerY * g->dx, NULL); /* FreeGrid.cpp:84 */
    /* GetResponse: INDEX(x, y, incDim) = x*(gx+1)+y, FDTD.cpp:76-77 */
    int idx = emitterX * (int)(unsigned)(g->gridSizeXf + 1) + emitterY;
    /* CalculateEFree: FreeGrid.cpp:96-110 */
    int numSamples = (int)(PV_DRY_GAIN_ANALYSIS_LENGTH * ((float)(int)g->fs)) +
                     (int)((1.f / PV_C) * (float)(int)g->fs);
    float efree = 0.f;
    for (int i = 0; i < numSamples; ++i) {
        float p = g->hist_pr[(size_t)i * g->ncell + idx];
        efree += p * p;
    }
    float r = (float)(emitterX - listenerX) * g->dx; /* FreeGrid.cpp:89-91 */
    efree *= r;
    pvo_grid_destroy(g);
    return efree;
}

/* Analyzer.cpp:139-328 */
static void encode_response(const PvoGrid* g, float efree, int serialIndex, int X, int Y, float lx, float lz,
                            float* res8, float* delay, unsigned char* valid, int offX, int offY) {
    const int numSamples = g->T;
    const size_t N = (size_t)g->ncell;
    const size_t cube = (size_t)X * (unsigned)(g->gridSizeXf + 1) + (size_t)Y; /* FDTD.cpp:76-77 */
    const float* hp = g->hist_pr + cube;
    const float* hx = g->hist_vx + cube;
    const float* hy = g->hist_vy + cube;
    const unsigned fs = g->fs;
    float* out = res8 + 8 * (size_t)serialIndex;
#define PR(j) ((j) < numSamples ? hp[(size_t)(j) * N] : 0.f)
#define VX(j) ((j) < numSamples ? hx[(size_t)(j) * N] : 0.f)
#define VY(j) ((j) < numSamples ? hy[(size_t)(j) * N] : 0.f)

    /* onset: Analyzer.cpp:146-165 */
    int onsetSample = 0;
    for (; onsetSample < numSamples; ++onsetSample) {
        float next = PR(onsetSample);
        if (fabsf(next) > PV_AUDIBLE_THRESHOLD_GAIN) break;
    }
    if (valid) valid[serialIndex] = 0;
    if (onsetSample < numSamples) {
        delay[serialIndex] = (float)onsetSample;
    } else {
        delay[serialIndex] = FLT_MAX;
        return;
    }

    /* dry: Analyzer.cpp:170-220 */
    int directGainSamples = (int)(PV_DRY_GAIN_ANALYSIS_LENGTH * (float)fs);
    int sourceDirSamples = (int)(PV_DRY_DIRECTION_ANALYSIS_LENGTH * (float)fs);
    int sourceDirEnd = onsetSample + sourceDirSamples;
    int directEnd = onsetSample + directGainSamples;
    float obstructionGain = 0.0f;
    float radx = 0.f, rady = 0.f;
    {
        float Edry = 0;
        int j = 0;
        for (; j < sourceDirEnd; ++j) {
            float p = PR(j);
            Edry += p * p;
            radx += p * VX(j);
            rady += p * VY(j);
        }
        for (; j < directEnd; ++j) {
            float p = PR(j);
            Edry += p * p;
        }
        const int listenerX = (int)(lx * (1.f / g->dx));
        const int listenerY = (int)(lz * (1.f / g->dx));
        float EfreePr = pvo_efree_per_r(efree, g->dx, listenerX, listenerY, X + offX, Y + offY);
        float E = (Edry / EfreePr);
        obstructionGain = sqrtf(E);
        float norm = sqrtf(radx * radx + rady * rady);
        norm = -1.0f / (norm > 0.0f ? norm : 1.0f);
        radx = norm * radx;
        rady = norm * rady;
    }
    out[0] = obstructionGain;
    out[6] = radx;
    out[7] = rady;

    /* lowpass: Analyzer.cpp:227-230 */
    float r = 1.0f / fmaxf(0.001f, obstructionGain);
    out[3] = -147.f + (18390.f) / (1.f + powf(r / 12.f, 0.8f));

    /* wet: Analyzer.cpp:235-247 */
    float wetEnergy = 0.0f;
    {
        const int wetGainSamples = (int)(PV_WET_GAIN_ANALYSIS_LENGTH * (float)fs);
        int end = directEnd + 1 + wetGainSamples;
        if (numSamples < end) end = numSamples;
        for (int j = directEnd + 1; j < end; j++) {
            float p = PR(j);
            wetEnergy += p * p;
        }
    }
    out[1] = sqrtf(wetEnergy / efree);

    /* rt60: Analyzer.cpp:282-327 */
    {
        int startingPoint = directEnd + 1;
        int endPoint = numSamples - (int)(PV_SCHROEDER_OFFSET_S * (float)fs);
        int regressN = endPoint - startingPoint;
        float rn = (float)regressN;
        float xmean = (rn - 1.0f) * 0.5f;
        float xsum = rn * xmean;
        float denominator = (1.0f / 12.0f) * rn * (rn * rn - 1.0f);
        float energyDecayCurve = 0.f, energyDecayCurveDB = 0.f, xysum = 0, ysum = 0;
        for (int i = numSamples - 1; i >= endPoint; --i) {
            float p = PR(i);
            energyDecayCurve += p * p;
        }
        for (int i = endPoint - 1; i >= startingPoint; --i) {
            float p = PR(i);
            energyDecayCurve += p * p;
            energyDecayCurveDB = 10.f * log10f(energyDecayCurve);
            float y_i = energyDecayCurveDB;
            int x_i = (i - startingPoint);
            xysum += y_i * (float)x_i;
            ysum += y_i;
        }
        float ymean = ysum / rn;
        float numerator = xysum - ymean * xsum - xmean * ysum + rn * xmean * ymean;
        float slopeDBperSample = numerator / denominator;
        float slopeDBperSec = slopeDBperSample * (float)fs;
        out[2] = -60.f / slopeDBperSec;
        /* SURVEY Q5 validity mask: onset + N_dry + 2 <= T - N_cut.  Outside it the reference may read
         * heap memory past the IR (Analyzer.cpp:191-195) and rt60 degenerates to inf/NaN. */
        if (valid) valid[serialIndex] = (directEnd + 2 <= endPoint) ? 1 : 0;
    }
#undef PR
#undef VX
#undef VY
}

/* Analyzer.cpp:332-337 */
static const int NB[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};

/* Analyzer.cpp:340-431 */
static void encode_listener_direction(const PvoGrid* g, int index, float lx, float lz, const float* res8,
                                      const float* delayMap, float* outx, float* outy, int offX, int offY) {
    const unsigned dimx = (unsigned)g->gx, dimy = (unsigned)g->gy;
    float loudness = res8[8 * (size_t)index + 0];
    int nextIndex = index;
    float delay = FLT_MAX;
    const float samplingRate = (float)g->fs;
    const float wavelength = PV_C / (float)g->res;
    const float thresholdDist = 0.3f * wavelength;

    while (delay > PV_DELAY_CLOSE_THRESHOLD && loudness < PV_DISTANCE_GAIN_THRESHOLD) {
        int r = (int)((unsigned)nextIndex / dimx), c = (int)((unsigned)nextIndex % dimx);
        float nextLoudness = 0.f;
        float nextDelay = FLT_MAX;
        for (int i = 0; i < 8; ++i) {
            int nr = r + NB[i][0], nc = c + NB[i][1];
            if (nr < 0 || nc < 0 || nr >= (int)dimx || nc >= (int)dimy) continue;
            int newPosIndex = (int)(nr * dimx + nc);
            float occ = res8[8 * (size_t)newPosIndex + 0];
            float d = delayMap[newPosIndex];
            /* (unsigned)delay == numSamples (Analyzer.cpp:372) is never true: delays are < T or FLT_MAX */
            if (occ == 0.f)
                continue;
            else if (d < nextDelay && occ > 0.f) {
                nextLoudness = occ;
                nextIndex = newPosIndex; /* overwritten even if the step is later rejected (Analyzer.cpp:377) */
                nextDelay = d;
            }
        }
        if (nextDelay == FLT_MAX || nextDelay >= delay) break;
        delay = nextDelay;
        loudness = nextLoudness;
        float geodesicDist = PV_C * nextDelay / samplingRate;
        int r2 = (int)((unsigned)nextIndex / dimx), c2 = (int)((unsigned)nextIndex % dimx);
        float ex = (float)(r2 + offX) * g->dx, ey = (float)(c2 + offY) * g->dx;
        float tx = ex - lx, ty = ey - lz;
        float euclideanDist = sqrtf((tx * tx) + (ty * ty));
        float distCheck = fabsf(geodesicDist - euclideanDist);
        if (distCheck < thresholdDist) break;
    }
    int r = (int)((unsigned)nextIndex / dimx), c = (int)((unsigned)nextIndex % dimx);
    float ex = (float)(r + offX) * g->dx, ey = (float)(c + offY) * g->dx;
    float ox = ex - lx, oy = ey - lz;
    float length = (ox * ox) + (oy * oy);
    if (length != 0.f) {
        length = sqrtf(length);
        ox /= length;
        oy /= length;
    }
    *outx = ox;
    *outy = oy;
}

/* Analyzer.cpp:48-104.
 * (offX, offY) = 0 is the reference.  A non-zero offset treats this grid as the window [offX, offX+gx] x
 * [offY, offY+gy] of a LARGER grid whose listener sits at (lx, lz) metres of the large grid: the impulse responses
 * are this grid's (an open field is translation-invariant), every piece of POSITION arithmetic -- the listener cell
 * and the cell coordinates in GetEFreePerR (FreeGrid.cpp:41-59), cellPos in EncodeListenerDirection
 * (Analyzer.cpp:395-398,415-417) -- uses the large grid's coordinates.  The 8-neighbour bounds test stays this
 * window's, so only cells whose walks stay inside the window are meaningful.  Used for BASELINE config 5 (8192^2). */
void pvo_analyze_at(const PvoGrid* g, float efree, float lx, float lz, int offX, int offY, float* res8, float* delay,
                    unsigned char* valid) {
    const int gridSize = g->gx * g->gy;
    const unsigned dimx = (unsigned)g->gx;
    for (int i = 0; i < gridSize; ++i) delay[i] = FLT_MAX;
    for (int s = 0; s < gridSize; ++s) {
        int X = (int)((unsigned)s / dimx), Y = (int)((unsigned)s % dimx); /* INDEX_TO_POS, stride gx */
        encode_response(g, efree, s, X, Y, lx, lz, res8, delay, valid, offX, offY);
    }
    for (int s = 0; s < gridSize; ++s) {
        float ox, oy;
        encode_listener_direction(g, s, lx, lz, res8, delay, &ox, &oy, offX, offY);
        res8[8 * (size_t)s + 4] = ox;
        res8[8 * (size_t)s + 5] = oy;
    }
}

void pvo_analyze(const PvoGrid* g, float efree, float lx, float lz, float* res8, float* delay,
                 unsigned char* valid) {
    pvo_analyze_at(g, efree, lx, lz, 0, 0, res8, delay, valid);
}

/* Analyzer.cpp:106-116 (keeps the reference's '>' test: SURVEY quirk Q6) */
int pvo_result_index(const PvoGrid* g, float ex, float ez) {
    unsigned posX = (unsigned)((ex + 0.f) / g->dx);
    unsigned posY = (unsigned)((ez + 0.f) / g->dx);
    if (posX > (unsigned)g->gx || posY > (unsigned)g->gy) return -1;
    return (int)(posX * (unsigned)g->gx + posY);
}

/* PlaneverbDSP/src/PvDSPContext.cpp:165-228 */
void pvo_find_gains(float rt60, float wet, float* a, float* b, float* c) {
    const float T1 = 0.5f, T2 = 1.0f, T3 = 3.0f, TSTAR = 0.1f;
    /* FindGainA :165-182 */
    if (rt60 > T2)
        *a = 0.f;
    else if (rt60 < T1)
        *a = 1.f;
    else {
        float term1 = powf(10.f, -3.f * TSTAR / T2);
        float term2 = powf(10.f, -3.f * TSTAR / rt60);
        float term3 = powf(10.f, -3.f * TSTAR / T1);
        *a = wet * (term1 - term2) / (term1 - term3);
    }
    /* FindGainB :184-209 */
    if (rt60 < T1)
        *b = 0.f;
    else {
        float term2 = powf(10.f, -3.f * TSTAR / rt60);
        if (rt60 > T2) {
            float term1 = powf(10.f, -3.f * TSTAR / T3);
            float term3 = powf(10.f, -3.f * TSTAR / T2);
            *b = wet * (term1 - term2) / (term1 - term3);
        } else {
            float term1 = powf(10.f, -3.f * TSTAR / T2);
            float term3 = powf(10.f, -3.f * TSTAR / T1);
            float aa = wet * (term1 - term2) / (term1 - term3);
            *b = wet - aa;
        }
    }
    /* FindGainC :211-228 */
    if (rt60 > T3)
        *c = 1.f;
    else if (rt60 < T2)
        *c = 0.f;
    else {
        float term1 = powf(10.f, -3.f * TSTAR / T3);
        float term2 = powf(10.f, -3.f * TSTAR / rt60);
        float term3 = powf(10.f, -3.f * TSTAR / T2);
        float aa = wet * (term1 - term2) / (term1 - term3);
        *c = wet - aa;
    }
}
