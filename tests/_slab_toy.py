"""A toy stand-in for api.SlabRank / api.SlabRoot (tests/test_dist_cpu.py): an integer-valued 3-point stencil along the
rows of a (rows x cols) field, advanced K steps per launch on slab + K halo rows (the trapezoid the real kernels compute),
with a per-cell "analysis" whose first row needs the history of the row above -- everything the exchange schedule of
planeverb_amd.dist_slabs moves, in exact arithmetic, so the decomposed result must EQUAL the single-domain one."""
import numpy as np

M = 8191  # values stay below 2^24: exact in float32


def step(u):
    """one step on an array of rows; the first and last row of the array lose validity (halo)"""
    v = u.copy()
    v[1:-1] = (u[:-2] + 2 * u[1:-1] + 3 * u[2:] + 1) % M
    return v


def whole_domain(NX, cols, T, K, src):
    """(final field, history [T, NX, cols]) of the undivided domain; rows outside are zero"""
    u = np.zeros((NX + 2 * K, cols), np.float64)
    hist = np.zeros((T, NX, cols), np.float64)
    for t in range(T):
        u[src[0] + K, src[1]] += t + 1
        u[:K] = 0
        u[K + NX:] = 0
        u = step(u)
        u[:K] = 0
        u[K + NX:] = 0
        hist[t] = u[K:K + NX]
    return u[K:K + NX], hist


def analysis(hist_rows, hist_above):
    """per cell: sum_t (p_t[r] - p_t[r-1]) mod M, the first row against `hist_above` [T, cols] (zeros for the domain's
    first row); 7 planes = that value + k"""
    up = np.concatenate([hist_above[:, None, :], hist_rows[:, :-1, :]], 1)
    a = (hist_rows - up).sum(0) % M
    return np.stack([a + k for k in range(7)]).astype(np.float32)


class ToySlab:
    def __init__(self, NX, cols, T, K, index, count, src):
        self.NX, self.cols, self.T, self.K, self.index, self.count, self.src = NX, cols, T, K, index, count, src
        self.r0 = NX * index // count
        self.r1 = NX * (index + 1) // count
        self.n = self.r1 - self.r0
        self.num_launches = -(-T // K)
        self.halo_floats = K * cols
        self.history_floats = T * cols
        self.above = np.zeros((T, cols), np.float64)

    def begin(self, listener):
        self.u = np.zeros((self.n + 2 * self.K, self.cols), np.float64)
        self.hist = np.zeros((self.T, self.n, self.cols), np.float64)

    def launch(self, li):
        K = self.K
        for t in range(li * K, min((li + 1) * K, self.T)):
            sr = self.src[0] - self.r0 + K
            if 0 <= sr < self.u.shape[0]:
                self.u[sr, self.src[1]] += t + 1
            # rows outside the whole domain stay zero (the real grid's guard band)
            if self.index == 0:
                self.u[:K] = 0
            if self.index == self.count - 1:
                self.u[K + self.n:] = 0
            self.u = step(self.u)
            if self.index == 0:
                self.u[:K] = 0
            if self.index == self.count - 1:
                self.u[K + self.n:] = 0
            self.hist[t] = self.u[K:K + self.n]
        # the halo rows are stale now (the trapezoid shrank to the slab's own rows): poison them, the exchange must refill
        if self.index > 0:
            self.u[:K] = -1e6
        if self.index < self.count - 1:
            self.u[K + self.n:] = -1e6

    def export_halo(self, side):
        K = self.K
        rows = self.u[K:2 * K] if side == 0 else self.u[self.n:self.n + K]
        return rows.astype(np.float32).ravel()

    def import_halo(self, side, buf):
        K = self.K
        rows = np.asarray(buf, np.float64).reshape(K, self.cols)
        if side == 0:
            self.u[:K] = rows
        else:
            self.u[K + self.n:] = rows

    def export_edge_history(self):
        return self.hist[:, -1, :].astype(np.float32).ravel()

    def import_above_history(self, buf):
        self.above = np.asarray(buf, np.float64).reshape(self.T, self.cols)

    def analyze(self):
        self.res = analysis(self.hist, self.above)

    def window_block(self):
        return np.array([self.r0, 0, self.n, self.cols], np.int32), self.res.ravel()


class ToyRoot:
    def __init__(self, NX, cols):
        self.maps = np.zeros((7, NX, cols), np.float32)
        self.finished = False

    def begin(self, listener):
        self.maps[:] = -1
        self.finished = False

    def import_block(self, info, data):
        r0, c0, nr, nc = (int(v) for v in info)
        self.maps[:, r0:r0 + nr, c0:c0 + nc] = np.asarray(data, np.float32).reshape(7, nr, nc)

    def finish(self):
        assert (self.maps >= 0).all(), "a block is missing"
        self.finished = True
