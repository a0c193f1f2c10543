// Problem geometry on the device: hull points, obstacle tables with exact collision thresholds, occupancy grid + coarse map,
// uniform box grid (config 5); Riccati weights and the per-sample S launch.  Fragment of engine.hip.
// Uniform grid over the bounding volume of the box obstacles (BASELINE.json config 5: 100k boxes).  Every
// box is registered in each cell it overlaps (closed intervals, one cell of slack), so "point inside some
// box" is decided from the point's own cell only -- the same boolean as the brute-force sweep.
static int build_box_grid(lqrrt_engine* e, const lqrrt_system_desc* sys) {
    const int O = sys->n_obstacles;
    const double* b = sys->obs;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, ext = 0.0;
    for (int o = 0; o < O; ++o)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], b[6 * o + d]); hi[d] = std::max(hi[d], b[6 * o + 3 + d]);
            ext = std::max(ext, b[6 * o + 3 + d] - b[6 * o + d]);
        }
    // cell edge: at least the largest box edge, and coarse enough for <= ~2M cells
    double vol = 1.0;
    for (int d = 0; d < 3; ++d) vol *= std::max(hi[d] - lo[d], 1e-9);
    double cell = std::max(ext, std::cbrt(vol / std::max(1, std::min(O * 2, 2000000))));
    if (!(cell > 0.0) || !std::isfinite(cell)) cell = 1.0;
    int dim[3];
    for (int d = 0; d < 3; ++d) dim[d] = std::max(1, (int)std::floor((hi[d] - lo[d]) / cell) + 1);
    const size_t ncell = (size_t)dim[0] * dim[1] * dim[2];
    auto cidx = [&](double v, int d) {
        int c = (int)std::floor((v - lo[d]) / cell);
        return std::min(std::max(c, 0), dim[d] - 1);
    };
    std::vector<int> count(ncell + 1, 0);
    auto for_cells = [&](int o, auto&& fn) {
        int c0[3], c1[3];
        for (int d = 0; d < 3; ++d) {
            c0[d] = std::max(cidx(b[6 * o + d], d) - 1, 0);          // one cell of slack on both sides
            c1[d] = std::min(cidx(b[6 * o + 3 + d], d) + 1, dim[d] - 1);
        }
        for (int i = c0[0]; i <= c1[0]; ++i)
            for (int j = c0[1]; j <= c1[1]; ++j)
                for (int k = c0[2]; k <= c1[2]; ++k) fn(((size_t)i * dim[1] + j) * dim[2] + k);
    };
    for (int o = 0; o < O; ++o) for_cells(o, [&](size_t c) { count[c + 1]++; });
    for (size_t c = 0; c < ncell; ++c) count[c + 1] += count[c];
    std::vector<int> items((size_t)count[ncell] + 1), fill(count.begin(), count.end() - 1);
    for (int o = 0; o < O; ++o) for_cells(o, [&](size_t c) { items[(size_t)fill[c]++] = o; });
    TRY(dalloc(&e->d_cell_start, ncell + 1));
    TRY(dalloc(&e->d_cell_items, items.size()));
    HIPCHK(hipMemcpy(e->d_cell_start, count.data(), sizeof(int) * (ncell + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_cell_items, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice));
    e->geo.cell_start = e->d_cell_start; e->geo.cell_items = e->d_cell_items;
    for (int d = 0; d < 3; ++d) { e->geo.glo[d] = lo[d]; e->geo.ghi[d] = hi[d]; e->geo.gdim[d] = dim[d]; }
    e->geo.gcell = cell;
    return 0;
}

// Problem geometry on the device: hull points, obstacle table (+ exact collision thresholds / box grid) and
// the optional occupancy grid.  Used at creation and by lqrrt_engine_set_geometry.
static int upload_geometry(lqrrt_engine* e, const lqrrt_system_desc* sys) {
    int rc = 0;
    auto up = [&](double** dst, const double* src, size_t cnt) -> int {
        TRY(dalloc(dst, cnt));
        if (cnt) HIPCHK(hipMemcpy(*dst, src, cnt * sizeof(double), hipMemcpyHostToDevice));
        return 0;
    };
    e->geo.V = sys->n_vertices; e->geo.O = sys->n_obstacles;
    for (int k = 0; k < 4; ++k) e->geo.hbb[k] = 0.0;
    e->geo.stride = sys->obs_stride > 0 ? sys->obs_stride : 3; e->geo.pad = 0;
    rc = up(&e->d_vps, sys->vps, (size_t)2 * sys->n_vertices);
    if (!rc) rc = up(&e->d_obs, sys->obs, (size_t)sys->n_obstacles * e->geo.stride);
    e->geo.vps = e->d_vps; e->geo.obs = e->d_obs; e->geo.oc = nullptr;
    if (!rc && e->geo.stride == 3) {
        // exact square-root-free collision thresholds + conservative reach radii (see systems.hpp Geo)
        double hull_r = 0.0;
        for (int v = 0; v < sys->n_vertices; ++v)
            hull_r = std::max(hull_r, std::sqrt(sys->vps[v] * sys->vps[v] + sys->vps[sys->n_vertices + v] * sys->vps[sys->n_vertices + v]));
        const double inflate = model_novice(sys->model) ? sys->params[18] : 0.0;
        std::vector<double> oc((size_t)4 * sys->n_obstacles + 4);
        for (int o = 0; o < sys->n_obstacles; ++o) {
            const double r = model_novice(sys->model) ? inflate + sys->obs[3 * o + 2] : sys->obs[3 * o + 2];
            oc[4 * o] = sys->obs[3 * o]; oc[4 * o + 1] = sys->obs[3 * o + 1];
            oc[4 * o + 2] = exact_sq_threshold(r);
            oc[4 * o + 3] = (r >= 0.0) ? r * (1.0 + 1e-9) + 1e-9 : -1e300;      // padded radius for the cull (never near if invalid)
        }
        double bb[4] = {0.0, 0.0, 0.0, 0.0};
        for (int v = 0; v < sys->n_vertices; ++v) {
            const double bx = sys->vps[v], by = sys->vps[sys->n_vertices + v];
            if (v == 0) { bb[0] = bb[1] = bx; bb[2] = bb[3] = by; }
            bb[0] = std::min(bb[0], bx); bb[1] = std::max(bb[1], bx);
            bb[2] = std::min(bb[2], by); bb[3] = std::max(bb[3], by);
        }
        for (int k = 0; k < 4; ++k) {
            const double pad = 1e-9 * (1.0 + std::fabs(bb[k]));
            e->geo.hbb[k] = (k & 1) ? bb[k] + pad : bb[k] - pad;
        }
        (void)hull_r;
        rc = up(&e->d_oc, oc.data(), (size_t)4 * sys->n_obstacles);
        e->geo.oc = e->d_oc;
    }
    e->geo.og = nullptr; e->geo.ogc = nullptr; e->geo.og_lds = 0;
    if (!rc && sys->ogrid) {
        if (sys->og_rows < 1 || sys->og_cols < 1 || !(sys->og_cpm > 0)) rc = fail(LQRRT_E_ARG, "bad occupancy grid");
        if (!rc) rc = dalloc(&e->d_og, (size_t)sys->og_rows * sys->og_cols);
        if (!rc && hipMemcpy(e->d_og, sys->ogrid, (size_t)sys->og_rows * sys->og_cols, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(LQRRT_E_HIP, "ogrid upload failed");
        e->geo.og = e->d_og; e->geo.og_rows = sys->og_rows; e->geo.og_cols = sys->og_cols;
        e->geo.og_ox = sys->og_origin[0]; e->geo.og_oy = sys->og_origin[1];
        e->geo.og_cpm = sys->og_cpm; e->geo.og_thr = sys->og_threshold;
        // coarse map for the conservative cull in grid_hits: block (R, C) = any occupied cell in its 8x8 cells,
        // "occupied" exactly as the sweep reads it: not (value < threshold)
        const int cr = (sys->og_rows + (1 << OG_COARSE_SHIFT) - 1) >> OG_COARSE_SHIFT;
        const int cc = (sys->og_cols + (1 << OG_COARSE_SHIFT) - 1) >> OG_COARSE_SHIFT;
        std::vector<unsigned char> coarse((size_t)cr * cc, 0);
        for (int r = 0; r < sys->og_rows; ++r)
            for (int c = 0; c < sys->og_cols; ++c)
                if (!((double)sys->ogrid[(size_t)r * sys->og_cols + c] < sys->og_threshold))
                    coarse[(size_t)(r >> OG_COARSE_SHIFT) * cc + (c >> OG_COARSE_SHIFT)] = 1;
        if (!rc) rc = dalloc(&e->d_ogc, coarse.size());
        if (!rc && hipMemcpy(e->d_ogc, coarse.data(), coarse.size(), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(LQRRT_E_HIP, "ogrid upload failed");
        e->geo.ogc = e->d_ogc; e->geo.ogc_rows = cr; e->geo.ogc_cols = cc;
        double hull_r = 0.0;
        for (int v = 0; v < sys->n_vertices; ++v)
            hull_r = std::max(hull_r, std::sqrt(sys->vps[v] * sys->vps[v] + sys->vps[sys->n_vertices + v] * sys->vps[sys->n_vertices + v]));
        e->geo.og_reach = hull_r * (1.0 + 1e-9) + 1e-9;
        double bb[4] = {0.0, 0.0, 0.0, 0.0};
        for (int v = 0; v < sys->n_vertices; ++v) {
            const double bx = sys->vps[v], by = sys->vps[sys->n_vertices + v];
            if (v == 0) { bb[0] = bb[1] = bx; bb[2] = bb[3] = by; }
            bb[0] = std::min(bb[0], bx); bb[1] = std::max(bb[1], bx);
            bb[2] = std::min(bb[2], by); bb[3] = std::max(bb[3], by);
        }
        for (int k = 0; k < 4; ++k) e->geo.og_bb[k] = bb[k];
        e->geo.og_lds = ((size_t)2 * sys->n_vertices * sizeof(double) <= (size_t)48 * 1024) ? 1 : 0;
    }
    e->geo.cell_start = nullptr; e->geo.cell_items = nullptr;
    if (!rc && e->geo.stride == 6 && sys->n_obstacles > 0) rc = build_box_grid(e, sys);
    return rc;
}

// Riccati systems: weights Q, R of the parameter block on the device (k_lqr_dare reads them from HBM)
static int upload_weights(lqrrt_engine* e) {
    if (!e->riccati) return 0;
    if (!e->d_QR) TRY(dalloc(&e->d_QR, (size_t)e->n * e->n + (size_t)e->m * e->m));
    HIPCHK(hipMemcpy(e->d_QR, e->P.p + riccati_q(e->model), sizeof(double) * e->n * e->n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->d_QR + e->n * e->n, e->P.p + riccati_r(e->model), sizeof(double) * e->m * e->m, hipMemcpyHostToDevice));
    return 0;
}

// S = lqr(x, 0)[0] for B states (Riccati systems): the cost-to-go matrix about each sample, planner.py:344-345
static int launch_sample_S(lqrrt_engine* e, const double* xs, int B, double* S_out, hipStream_t st) {
    if (B <= 0) return 0;
    if (!e->has_res) return fail(LQRRT_E_STATE, "set_resolution first (dt)");
    DISPATCH(e, hipLaunchKernelGGL((k_lqr_dare<S>), dim3(B), dim3(64), 0, st, e->P, xs, (const double*)nullptr, B, e->d_QR,
                                   e->d_QR + e->n * e->n, e->res.dt, e->P.p[riccati_eps(e->model)], 64, 1e-14, S_out, (double*)nullptr,
                                   (double*)nullptr, (double*)nullptr, (int*)nullptr));
    HIPCHK(hipGetLastError());
    return 0;
}

static void free_geometry(lqrrt_engine* e) {
    void** ptrs[] = {(void**)&e->d_vps, (void**)&e->d_obs, (void**)&e->d_oc, (void**)&e->d_og, (void**)&e->d_ogc, (void**)&e->d_cell_start, (void**)&e->d_cell_items};
    for (void** p : ptrs) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}
