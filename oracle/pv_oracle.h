/* oracle/pv_oracle.h -- TEST INFRASTRUCTURE ONLY (checker, never shipped, never measured as the product).
 *
 * Plain-C restatement of the reference hot path.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/).  All arithmetic is float32 in the
 * reference's operation order; build with -ffp-contract=off and without fast-math (oracle/Makefile).
 *
 * Pinning: tests/test_oracle_vs_ref.py checks this file bit-for-bit against oracle/_ref/libpvref.so (the
 * unmodified reference compiled from /root/reference) and against tests/golden/ fixtures generated from it.
 *
 * Layout: the reference's (gx+1)x(gy+1) cell array, linear index i = x*S + y with S = gy+1 (FDTD.cpp:99).
 * Histories are SoA [t][cell] (the reference keeps one std::vector<Cell> per cell; same values).
 */
#ifndef PV_ORACLE_H
#define PV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PvoGrid {
    int res;          /* gridResolution */
    float sizeX, sizeY; /* gridSizeInMeters */
    float dx, dt;
    unsigned fs;      /* sampling rate */
    int T;            /* response length */
    float gridSizeXf, gridSizeYf; /* the reference's float m_gridSize */
    int gx, gy;       /* (int)m_gridSize */
    int ncell;        /* (gx+1)*(gy+1) */
    short* b;         /* beta plane, ncell */
    float* R;         /* absorption plane, ncell */
    float* pulse;     /* T */
    float* hist_pr;   /* T*ncell, filled by pvo_fdtd */
    float* hist_vx;
    float* hist_vy;
} PvoGrid;

/* Grid.cpp:390-396 */
void pvo_grid_params(int res, float* dx, float* dt, unsigned* fs);
/* Grid.cpp:55 + PvTypes.h:101 */
int pvo_response_length(unsigned fs);
/* Grid.cpp:12-27 */
void pvo_gaussian_pulse(int res, unsigned fs, float* out, int n);

/* Grid.cpp:30-117 (no IR cube unless with_history) */
PvoGrid* pvo_grid_create(float sizeX, float sizeY, int res, int with_history);
void pvo_grid_destroy(PvoGrid* g);
/* Grid.cpp:136-144,229-246 ; a5 = {posX,posY,width,height,absorption} */
void pvo_add_aabb(PvoGrid* g, const float* a5);
/* Grid.cpp:249-296 */
void pvo_remove_aabb(PvoGrid* g, const float* a5);

/* FDTD.cpp:97-99 */
void pvo_listener_cell(const PvoGrid* g, float lx, float lz, int* cx, int* cy);
/* FDTD.cpp:87-236.  Needs histories.  If fields!=NULL, the final pr,vx,vy (3*ncell) are copied out. */
void pvo_fdtd(PvoGrid* g, float lx, float lz, float* fields);

/* FreeGrid.cpp:6-34,71-110 : full extra run on an empty grid of the same config */
float pvo_free_energy(float sizeX, float sizeY, int res);
/* FreeGrid.cpp:41-59 */
float pvo_efree_per_r(float efree, float dx, int lX, int lY, int eX, int eY);

/* Analyzer.cpp:48-104,139-328,340-431.  res8: gx*gy*8 (must be zero-initialised by a fresh context,
 * PvContext.cpp:132 -- cells without onset are left untouched, Analyzer.cpp:160-165), delay: gx*gy,
 * valid: gx*gy (1 where the reference reads no memory past the IR: SURVEY Q5; may be NULL). */
void pvo_analyze(const PvoGrid* g, float efree, float lx, float lz, float* res8, float* delay,
                 unsigned char* valid);
/* The same with this grid taken as the window at cell offset (offX, offY) of a larger open grid (see pv_oracle.c);
 * pvo_analyze == pvo_analyze_at(..., 0, 0, ...), which is the form pinned against the compiled reference. */
void pvo_analyze_at(const PvoGrid* g, float efree, float lx, float lz, int offX, int offY, float* res8, float* delay,
                    unsigned char* valid);

/* Analyzer.cpp:106-116 : returns result-map index or -1 */
int pvo_result_index(const PvoGrid* g, float ex, float ez);

/* PlaneverbDSP/src/PvDSPContext.cpp:165-228 : reverb-bus split ("RT60 bucket") */
void pvo_find_gains(float rt60, float wet, float* a, float* b, float* c);

#ifdef __cplusplus
}
#endif
#endif
