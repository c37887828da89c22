// slice_loop.hip — one (measurement, configuration) pair on the GPU: atoms of (k, j), projected potential, the six LDS row passes
// of the fused slice loop resp. rocFFT + point-wise kernels, hipGraph capture / replay of the loop, incoming wave,
// exit-wave post-processing and the detector chain (src/crystalMaker.cu:324-373, 507-536, 579-613, 700-718;
// src/multisliceSimulation.cu:538-622).  Split from engine.hip in round 5; shared declarations: engine_impl.h.
#include "engine_impl.h"

namespace fdes_engine {

// tiltCoordinates, src/crystalMaker.cu:427-454.  cos/sin on the host, as the reference.
int tilt_coordinates(fdes_plan* pl, float* xyz, float t_0, float t_1, float t_2)
{
    hipStream_t st = pl->ctx->stream;
    if (fabsf(t_2) > FLT_EPSILON) HIPCHK(pl->ctx, geom_srot(xyz, pl->nAt, 0, 1, cosf(t_2), -sinf(t_2), st));
    if (fabsf(t_1) > FLT_EPSILON) HIPCHK(pl->ctx, geom_srot(xyz, pl->nAt, 0, 2, cosf(t_1), -sinf(t_1), st));
    if (fabsf(t_0) > FLT_EPSILON) HIPCHK(pl->ctx, geom_srot(xyz, pl->nAt, 1, 2, cosf(t_0), -sinf(t_0), st));
    return FDES_OK;
}

// src/crystalMaker.cu:330-331
int ensure_tilt(fdes_plan* pl, int k)
{
    if (pl->cur_k == k) return FDES_OK;
    HIPCHK(pl->ctx, hipMemcpyAsync(pl->xyzK_d, pl->xyzTO_d, sizeof(float) * 3 * (size_t)pl->nAt, hipMemcpyDeviceToDevice, pl->ctx->stream));
    RC(tilt_coordinates(pl, pl->xyzK_d, pl->p.tiltspec[2 * k], pl->p.tiltspec[2 * k + 1], 0.f));
    pl->cur_k = k;
    return FDES_OK;
}

// Option skip_empty: which slices of the configurations (ks[g], js[g]), g < n, hold atoms -> pl->seg_h (slice q counts as
// occupied when it is occupied in ANY of them: a gang skips a slice only when it is empty in every member).  Asked on the
// plan's query stream from the constant tilt-offset coordinates (geom_slice_occupancy recomputes tilt, jitter and the
// binning's slice test), so the one host wait per question covers a few microseconds of work of its own and NOT the slice
// loops queued on the lane's stream (until round 3 the question read the binning's segment table behind them).
// pl->gseg[g] receives member g's own table when n > 1.
int empty_query(fdes_plan* pl, int n, const int* ks, const int* js)
{
    fdes_ctx* c = pl->ctx;
    const int m3 = pl->p.m3, nZ = pl->nZ;
    const int cap = pl->gang > 1 ? pl->gang : 1;
    if (n > cap) return FDES_EINVAL;
    if (!pl->qs) {
        DeviceGuard guard(c->device); // stream creation vs a capture in another thread of this device
        int least = 0, greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIPCHK(c, hipStreamCreateWithPriority(&pl->qs, hipStreamNonBlocking, greatest)); // a queue of its own, ahead of the lanes' kernels
        RC(dmalloc(c, &pl->slice_occ_d, (size_t)cap * m3));
        HIPCHK(c, hipHostMalloc((void**)&pl->slice_occ_h, sizeof(int) * (size_t)cap * m3, hipHostMallocDefault));
    }
    BinGeom g{pl->p.m1, pl->p.m2, m3, nZ, pl->p.d1, pl->p.d2, pl->p.d3};
    for (int q = 0; q < n; q++)
        HIPCHK(c, geom_slice_occupancy(pl->slice_occ_d + (size_t)q * m3, pl->xyzTO_d, pl->dwf_d, pl->nAt, g, pl->p.tiltspec[2 * ks[q]], pl->p.tiltspec[2 * ks[q] + 1],
                                       pl->p.frPh > 0, owner_ctx(pl)->seed, ks[q], js[q], pl->qs));
    HIPCHK(c, hipMemcpyAsync(pl->slice_occ_h, pl->slice_occ_d, sizeof(int) * (size_t)n * m3, hipMemcpyDeviceToHost, pl->qs));
    HIPCHK(c, hipStreamSynchronize(pl->qs));
    const size_t len = (size_t)m3 * nZ + 1;
    auto table = [&](std::vector<int>& t, auto occupied) { // monotone, one step per occupied slice
        t.assign(len, 0);
        int cum = 0;
        for (int q = 0; q < m3; q++) {
            for (int z = 0; z < nZ; z++) t[(size_t)q * nZ + z] = cum;
            cum += occupied(q) ? 1 : 0;
        }
        t[len - 1] = cum;
    };
    if (n > 1)
        for (int q = 0; q < n; q++) {
            if (pl->gseg.size() < (size_t)n) pl->gseg.resize((size_t)n);
            table(pl->gseg[(size_t)q], [&](int s) { return pl->slice_occ_h[(size_t)q * m3 + s] != 0; });
        }
    table(pl->seg_h, [&](int s) { for (int q = 0; q < n; q++) if (pl->slice_occ_h[(size_t)q * m3 + s]) return true; return false; });
    return FDES_OK;
}

// src/crystalMaker.cu:335-337 + the per-configuration (slice, species) binning
int config_atoms(fdes_plan* pl, int k, int j, bool query, float* xyz, AtomBins* bins_p) // xyz / bins_p: a gang member's coordinates and binning buffers
{
    float* const xyzFP = xyz ? xyz : pl->xyzFP_d;
    AtomBins& bins = bins_p ? *bins_p : pl->bins;
    fdes_ctx* c = pl->ctx;
    RC(ensure_tilt(pl, k));
    if (pl->p.frPh > 0)
        HIPCHK(c, geom_jitter(xyzFP, pl->xyzK_d, pl->dwf_d, pl->nAt, owner_ctx(pl)->seed, k, j, c->stream));
    else
        HIPCHK(c, hipMemcpyAsync(xyzFP, pl->xyzK_d, sizeof(float) * 3 * (size_t)pl->nAt, hipMemcpyDeviceToDevice, c->stream));
    BinGeom g{pl->p.m1, pl->p.m2, pl->p.m3, pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3};
    HIPCHK(c, geom_bin_atoms(xyzFP, pl->spec_d, pl->occ_d, pl->nAt, g, bins, pl->fused || owner_ctx(pl)->deterministic, c->stream));
    if (!query) return FDES_OK; // (a gang asks once for all its members, gang_flush)
    fdes_plan* tp = pl->top ? pl->top : pl;
    constexpr int kDenseAfter = 8, kDenseRecheck = 64;
    bool ask = pl->fused && owner_ctx(pl)->skip_empty;
    if (ask && tp->dense_streak >= kDenseAfter && (tp->cfg_seen % kDenseRecheck) != 0) ask = false;
    tp->cfg_seen++;
    if (ask) {
        // which slices hold atoms decides the launch sequence
        const int kk = k, jj = j;
        RC(empty_query(pl, 1, &kk, &jj));
        tp->empty_queries++;
        bool any_empty = false;
        for (int q = 0; q < pl->p.m3 && !any_empty; q++) any_empty = pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ];
        tp->dense_streak = any_empty ? 0 : tp->dense_streak + 1;
    } else {
        pl->seg_h.clear();
    }
    return FDES_OK;
}

// bandwidthLimit, src/multisliceSimulation.cu:552-560
int bandwidth_limit(fdes_plan* pl, float2* f)
{
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, fft_exec(pl,f, false, c->stream));
    HIPCHK(c, k_mask_scale(f, pl->p.m1, pl->p.m2, 1.f / ((float)pl->m12), c->stream));
    HIPCHK(c, fft_exec(pl,f, true, c->stream));
    return FDES_OK;
}

// phaseGrating, src/crystalMaker.cu:507-536 -> VH (real space potential of sub-slice s)
int phase_grating(fdes_plan* pl, const float* xyz, const BinGeom& g, int s)
{
    fdes_ctx* c = pl->ctx;
    KP kp = pl->kp;
    kp.m3 = g.m3;
    kp.d3 = g.d3;
    for (int z = 0; z < pl->nZ; z++) {
        if (owner_ctx(pl)->deterministic && geom_deposit_tile_fits(g.m1)) HIPCHK(c, geom_deposit_tile(pl->D, pl->bins, s * pl->nZ + z, -1, true, pl->p.imPot, g, c->stream));
        else
        HIPCHK(c, geom_deposit(pl->D, xyz, pl->occ_d, pl->bins, s * pl->nZ + z, g, pl->p.imPot, pl->deposit_blocks, c->stream));
        HIPCHK(c, fft_exec(pl,pl->D, false, c->stream));
        HIPCHK(c, k_filter_accum(pl->VH, pl->D, kp, pl->kz[z], z == 0, c->stream));
    }
    HIPCHK(c, fft_exec(pl,pl->VH, true, c->stream));
    return FDES_OK;
}

// Packed potential of the slice pair (s0, s0 + 1) on the generic path: W = V_s0 + i V_(s0+1) in VH from one forward
// transform per species and one inverse transform for two slices (the deposits are real, the filter is real and
// even), with the filter read from the per-plan table instead of being re-evaluated (3 expf + 2 sinf per pixel).
int phase_grating_pair(fdes_plan* pl, const float* xyz, const BinGeom& g, int s0)
{
    fdes_ctx* c = pl->ctx;
    for (int z = 0; z < pl->nZ; z++) {
        const int k0 = s0 * pl->nZ + z, k1 = (s0 + 1 < g.m3) ? (s0 + 1) * pl->nZ + z : -1;
        if (owner_ctx(pl)->deterministic && geom_deposit_tile_fits(g.m1)) HIPCHK(c, geom_deposit_tile(pl->D, pl->bins, k0, k1, false, 0.f, g, c->stream));
        else
        HIPCHK(c, geom_deposit_pair(pl->D, xyz, pl->occ_d, pl->bins, k0, k1, g, pl->deposit_blocks, c->stream));
        HIPCHK(c, fft_exec(pl, pl->D, false, c->stream));
        HIPCHK(c, k_filter_accum_tab(pl->VH, pl->D, pl->GT + (size_t)z * pl->m12, pl->m12, z == 0, c->stream));
    }
    HIPCHK(c, fft_exec(pl, pl->VH, true, c->stream));
    return FDES_OK;
}

// forwardPropagation, src/multisliceSimulation.cu:538-549 (V in VH; comp >= 0: component of the packed pair potential)
int forward_propagation(fdes_plan* pl, int comp)
{
    fdes_ctx* c = pl->ctx;
    if (comp >= 0) HIPCHK(c, k_transmit_comp(pl->T, pl->VH, pl->m12, comp, pl->p.imPot, c->stream));
    else HIPCHK(c, k_transmit(pl->T, pl->VH, pl->m12, c->stream));
    RC(bandwidth_limit(pl, pl->T));
    HIPCHK(c, k_mul(pl->PSI, pl->T, pl->PSI, pl->m12, c->stream));      // multiplyElementwise(t, psi)
    HIPCHK(c, fft_exec(pl,pl->PSI, false, c->stream));                  // convolveWithFrProp
    HIPCHK(c, k_mul(pl->PSI, pl->PSI, pl->P, pl->m12, c->stream));
    HIPCHK(c, fft_exec(pl,pl->PSI, true, c->stream));
    return FDES_OK;
}

// ---- fused slice loop: six LDS row passes per slice, no stand-alone point-wise kernel.
//   P1  D[y][x]        -FFT_x->                          A_z[kx][y]      (per species; clears D)
//   P2  A_z[kx][y]     -FFT_y, * G_z, sum_z, IFFT_y->    B[y][kx]
//   P3  B[y][kx]       -IFFT_x, exp(iV), FFT_x->         C[kx][y]
//   P4  C[kx][y]       -FFT_y, band limit / m12, IFFT_y-> E[y][kx]
//   P5  E, PSIH[y][kx] -IFFT_x both, t * psi, FFT_x->    F[kx][y]
//   P6  F[kx][y]       -FFT_y, * P, IFFT_y->             PSIH[y][kx]
// pass over the rows of an "N" grid ([y][kx], row length m1) with a transposed store into a "T" grid, and the reverse;
// callers whose input or output is a dense natural grid (PSI, T, user buffers) override the pitch with 0
PassArgs pass_x(fdes_plan* pl) { PassArgs a; a.tw0 = pl->fft->tw0x; a.tw1 = pl->fft->tw1x; a.jit = pl->fft->jit_x; a.tile_rows = pl->fft->rows_x; a.nrows = pl->p.m2; a.wg = pl->wg; a.pitch_in = pl->pitchN; a.pitch_out = pl->pitchT; a.walk = owner_ctx(pl)->walk; a.stagger = owner_ctx(pl)->stagger; return a; }
PassArgs pass_y(fdes_plan* pl) { PassArgs a; a.tw0 = pl->fft->tw0y; a.tw1 = pl->fft->tw1y; a.jit = pl->fft->jit_y; a.tile_rows = pl->fft->rows_y; a.nrows = pl->p.m1; a.wg = pl->wg; a.pitch_in = pl->pitchT; a.pitch_out = pl->pitchN; a.walk = owner_ctx(pl)->walk; a.stagger = owner_ctx(pl)->stagger; return a; }

// stream of the potential / transmission passes
hipStream_t vstream(fdes_plan* pl) { return (pl->split && !pl->tap_mode) ? pl->vs : pl->ctx->stream; }

// gang (fdes_plan::gn members in one launch, grid z = member): element strides of the operands between members
void gang_strides(const fdes_plan* pl, PassArgs& a, size_t in0, size_t out, size_t out2, size_t in1)
{
    if (pl->gn <= 1) return;
    a.nbatch = pl->gn;
    a.bstride_in0 = in0; a.bstride_out = out; a.bstride_out2 = out2; a.bstride_in1 = in1;
}

// roofline probe (bench.py): every probe_stride-th launch of the pass class the context's probe_pass names (1 = P1', 2 = P2,
// 3 = P3, 4 = P4, 5 = P5, 6 = P6) is bracketed by the start / stop events of the dispatch itself (hipExtLaunchKernelGGL)
int probe_bracket(fdes_plan* pl, PassArgs& a, int cls)
{
    fdes_ctx* c = pl->ctx;
    const fdes_ctx* oc = owner_ctx(pl);
    if (pl->capturing || oc->probe_stride <= 0 || oc->probe_pass != cls) return FDES_OK;
    if ((pl->fft_calls++ % (uint64_t)oc->probe_stride) != 0) return FDES_OK;
    if (pl->probe_used == pl->probe.size()) {
        EvPair e{};
        HIPCHK(c, hipEventCreate(&e.a));
        HIPCHK(c, hipEventCreate(&e.b));
        pl->probe.push_back(e);
    }
    EvPair* ev = &pl->probe[pl->probe_used++];
    a.ev_start = ev->a;
    a.ev_stop = ev->b;
    return FDES_OK;
}

// Potential of the slice PAIR (s0, s0 + 1), s0 even: W = V_s0 + i V_(s0+1) (the deposits are real and the filter
// G is real and even, so one complex transform carries two slices).  P1' builds the x-spectra of the deposit rows
// straight from the sorted atom records (no deposit grid), P2 applies the filter in (kx, ky) and sums the species.
int fused_potential_pair(fdes_plan* pl, int s0)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    {   // one launch, grid.y = species
        PassArgs a = pass_x(pl);
        a.out = pl->A;
        a.nspecies = pl->nZ; a.species_stride = pl->gsz;
        a.recs = pl->bins.recs_sorted; a.rowstart = pl->bins.rowstart;
        a.q0 = s0 * pl->nZ;
        a.q1 = (s0 + 1 < pl->p.m3) ? (s0 + 1) * pl->nZ : -1;
        gang_strides(pl, a, 0, pl->gsz * (size_t)pl->nZ);
        if (pl->gn > 1) { // every member deposits the same slice pair from its own records
            a.bstride_recs = pl->recs_stride; a.bstride_rowstart = pl->rowstart_stride;
            for (int g = 0; g < pl->gn; g++) { a.zq0[g] = a.q0; a.zq1[g] = a.q1; }
        }
        RC(probe_bracket(pl, a, 1));
        HIPCHK(c, lds_pass(m1, XF_FWD, MID_ATOMS, XF_NONE, true, a, vstream(pl)));
    }
    PassArgs b = pass_y(pl);
    b.in0 = pl->A; b.gtab = pl->GT; b.out = pl->B; b.nspecies = pl->nZ; b.species_stride = pl->gsz;
    gang_strides(pl, b, pl->gsz * (size_t)pl->nZ, pl->gsz);
    RC(probe_bracket(pl, b, 2));
    HIPCHK(c, lds_pass(m2, XF_FWD, pl->nZ == 1 ? MID_GTAB : MID_GTABN, XF_INV, true, b, vstream(pl)));
    return FDES_OK;
}

// separable table of P^n (n >= 2) for a run of empty slices: px^n[m1] | py^n[m2]
int propagator_pow(fdes_plan* pl, int n, float2** out)
{
    fdes_ctx* c = pl->ctx;
    const size_t len = (size_t)pl->p.m1 + (size_t)pl->p.m2;
    if (pl->capture_pow) {
        // inside a graph capture the table belongs to that graph: its build kernel is one of the nodes, so every replay
        // refreshes it and nothing depends on what other patterns did to a shared cache in between
        for (auto& e : *pl->capture_pow)
            if (e.first == n) { *out = e.second; return FDES_OK; }
        float2* t = nullptr;
        {
            RelaxCapture relax; // hipMalloc from the capturing thread
            RC(dmalloc(c, &t, len));
        }
        pl->capture_pow->push_back({n, t});
        HIPCHK(c, k_build_propagator_1d(t, t + pl->p.m1, pl->kp, n, c->stream));
        *out = t;
        return FDES_OK;
    }
    for (auto& e : pl->pow_tabs)
        if (e.n == n) { e.used = ++pl->pow_tick; *out = e.tab; return FDES_OK; }
    DeviceGuard guard(c->device); // hipMalloc vs a capture in another thread of this device
    float2* tab = nullptr;
    if (pl->pow_tabs.size() >= 16) {
        size_t lru = 0;
        for (size_t i = 1; i < pl->pow_tabs.size(); i++) if (pl->pow_tabs[i].used < pl->pow_tabs[lru].used) lru = i;
        tab = pl->pow_tabs[lru].tab; // stream order makes the overwrite safe: its last reader was enqueued earlier
        pl->pow_tabs.erase(pl->pow_tabs.begin() + (long)lru);
    } else {
        RC(dmalloc(c, &tab, len));
    }
    HIPCHK(c, k_build_propagator_1d(tab, tab + pl->p.m1, pl->kp, n, c->stream));
    pl->pow_tabs.push_back({n, tab, ++pl->pow_tick});
    *out = tab;
    return FDES_OK;
}

// a run of slices without atoms, starting at s: V = 0, t = BL(1) = 1; psi <- F^-1[P^n F[psi]] as one Fresnel step.
// PSIH is [y][kx]: a transposing copy gives the y-pass its rows, then the usual propagator pass.
int fused_empty_run(fdes_plan* pl, int s, int nslices, int* consumed)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0;
    auto empty = [&](int q) { return q >= pl->p.m3 || pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ]; };
    PassArgs a5 = pass_x(pl);
    a5.in0 = pl->PSIH; a5.out = pl->F;
    a5.scale = (float)m1; // P5 hands m1 * FFT_x(t psi) to P6 (unnormalised x round trip); exact power of two
    // (an incoming wave that is not band-limited in kx needs no special case here: the dead kx rows this copy drops are
    // zeroed by the masked propagator whatever they held)
    a5.band = band; a5.skip_dead_loads = bs; a5.skip_dead_stores = bs;
    gang_strides(pl, a5, pl->gsz, pl->gsz);
    HIPCHK(c, lds_pass(m1, XF_NONE, MID_SCALE, XF_NONE, true, a5, c->stream));
    int run = 1;
    while (s + run < nslices && empty(s + run)) run++;
    float2* ptab = pl->PT;
    if (run > 1) RC(propagator_pow(pl, run, &ptab));
    PassArgs a6 = pass_y(pl);
    a6.in0 = pl->F; a6.prow = ptab; a6.pcol = ptab + m1; a6.mindim = md; a6.out = pl->PSIH;
    a6.band = band; a6.live_rows_only = bs;
    gang_strides(pl, a6, pl->gsz, pl->gsz);
    HIPCHK(c, lds_pass(m2, XF_FWD, MID_PTAB, XF_INV, true, a6, c->stream));
    pl->slices_skipped += (int64_t)run * pl->gn;
    *consumed = run;
    return FDES_OK;
}

// the wave's two passes of slice s: P5 (t psi from the band-limited transmission spectrum E and psi-hat) and P6
// (Fresnel propagator); ei >= 0: the split loop's "E consumed" event of that buffer is recorded behind P5
int fused_wave_step(fdes_plan* pl, int s, const float2* E, int ei)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0;
    PassArgs a5 = pass_x(pl);
    a5.in0 = E; a5.in1 = pl->PSIH; a5.out = pl->F;
    // Dead kx columns: E's are never written by P4 (they may hold the pair potential's stale values: B aliases E), so
    // they are always skipped; psi-hat's are exact zeros after any masked propagator, but the FIRST product of a
    // configuration sees the incoming wave, which the reference multiplies by t in full (src/multisliceSimulation.cu:546)
    // and which is not band-limited in kx when a tilted CBED probe leaves the band (:583-590) - then all of it is read.
    a5.band = band; a5.skip_dead_loads = bs ? ((s == 0 && !pl->wave_bl) ? 1 : 3) : 0; a5.skip_dead_stores = bs;
    gang_strides(pl, a5, pl->gsz, pl->gsz, 0, pl->gsz);
    RC(probe_bracket(pl, a5, 5));
    HIPCHK(c, lds_pass(m1, XF_INV, MID_MULPSI, XF_FWD, true, a5, c->stream));
    if (ei >= 0) {
        HIPCHK(c, hipEventRecord(pl->evP5[ei], c->stream));
        pl->p5_seen[ei] = true;
    }
    PassArgs a6 = pass_y(pl);
    a6.in0 = pl->F; a6.prow = pl->PT; a6.pcol = pl->PT + m1; a6.mindim = md; a6.out = pl->PSIH;
    a6.band = band; a6.live_rows_only = bs;
    gang_strides(pl, a6, pl->gsz, pl->gsz);
    RC(probe_bracket(pl, a6, 6));
    HIPCHK(c, lds_pass(m2, XF_FWD, MID_PTAB, XF_INV, true, a6, c->stream));
    return FDES_OK;
}

// one slice of the fused loop; *consumed = slices advanced (a run of empty slices is one Fresnel step with P^n)
int fused_slice(fdes_plan* pl, int s, int nslices, int* consumed)
{
    *consumed = 1;
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2;
    // 2/3 band limit: rows/columns whose own frequency index already fails 9 i^2 <= mindim^2 are exact zeros after
    // P4 (mask) and P6 (masked propagator): P4/P6 run only their live row groups, P3/P5 do not store the rows those
    // never read, P5 does not load the columns they never write (pre-zeroed at plan creation).
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0; // the column classes of the passes assume the band of a square grid
    auto empty = [&](int q) { return q >= pl->p.m3 || pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ]; };
    const bool have_seg = !pl->seg_h.empty();
    if (have_seg && empty(s)) return fused_empty_run(pl, s, nslices, consumed);
    // the pair's potential and both transmission functions are built at the pair's first non-empty slice:
    // C <- F_x[t_s0], C2 <- F_x[t_(s0+1)] from one read and one inverse transform of W = V_s0 + i V_(s0+1)
    if ((s & 1) == 0 || (have_seg && empty(s - 1))) {
        RC(fused_potential_pair(pl, s & ~1));
        PassArgs a3 = pass_x(pl);
        a3.in0 = pl->B; a3.out = pl->C; a3.out2 = pl->C2; a3.scale = pl->p.imPot;
        a3.band = band; a3.skip_dead_stores = bs;
        gang_strides(pl, a3, pl->gsz, pl->gsz, pl->gsz);
        RC(probe_bracket(pl, a3, 3));
        HIPCHK(c, lds_pass(m1, XF_INV, MID_EXPIV_PAIR, XF_FWD, true, a3, vstream(pl)));
    }
    const bool split = pl->split && !pl->tap_mode;
    const int ei = s & 1;
    PassArgs a4 = pass_y(pl);
    a4.in0 = (s & 1) ? pl->C2 : pl->C; a4.out = pl->Eb[ei]; a4.scale = 1.f / ((float)pl->m12); a4.mindim = md;
    a4.band = band; a4.live_rows_only = bs;
    gang_strides(pl, a4, pl->gsz, pl->gsz);
    if (split && pl->p5_seen[ei]) HIPCHK(c, hipStreamWaitEvent(pl->vs, pl->evP5[ei], 0)); // the wave chain has consumed this buffer
    RC(probe_bracket(pl, a4, 4));
    HIPCHK(c, lds_pass(m2, XF_FWD, MID_MASK, XF_INV, true, a4, vstream(pl)));
    if (split) {
        HIPCHK(c, hipEventRecord(pl->evE[ei], pl->vs));
        HIPCHK(c, hipStreamWaitEvent(c->stream, pl->evE[ei], 0));
    }
    return fused_wave_step(pl, s, pl->Eb[ei], split ? ei : -1);
}

// The slices [0, nslices) of one configuration with the potential chain in BATCHES (plans with nb > 1): the potential does
// not depend on the wave (src/crystalMaker.cu:339-343: phaseGrating takes the atoms and the slice index only), so P1',
// P2, P3 of nb slice pairs and P4 of their 2 nb slices are one launch each (grid.z = pair resp. slice) on the potential
// stream - a single slice's rows cannot fill the chip at 1024^2 and below, and a single image has no second
// configuration to run beside it - while the wave stream runs P5 / P6 of the previous batch.  Empty slices take no
// part in the batch (skip_empty); the wave chain handles their runs as fused_slice does.
int batched_loop(fdes_plan* pl, int nslices)
{
    fdes_ctx* c = pl->ctx;
    const int m1 = pl->p.m1, m2 = pl->p.m2, nZ = pl->nZ, nb = pl->nb;
    const int md = m1 < m2 ? m1 : m2, band = md * md;
    const int bs = (owner_ctx(pl)->band_skip && m1 == m2) ? 1 : 0;
    const bool have_seg = !pl->seg_h.empty();
    auto empty = [&](int q) { return q >= pl->p.m3 || pl->seg_h[(size_t)(q + 1) * nZ] == pl->seg_h[(size_t)q * nZ]; };
    bool used[2] = {false, false};
    int sw = 0; // next slice of the wave chain
    int set = 0;
    for (int s0 = 0; s0 < nslices; s0 += 2 * nb, set ^= 1) {
        const int s1 = (s0 + 2 * nb < nslices) ? s0 + 2 * nb : nslices;
        // ---- potential chain of the batch
        int np = 0, ns = 0, eidx[16], zq0[16], zq1[16], zin[16];
        for (int i = 0; i < 16; i++) eidx[i] = -1;
        for (int sp = s0; sp < s1; sp += 2) {
            const bool e0 = have_seg && empty(sp), e1 = (sp + 1 >= s1) || (have_seg && empty(sp + 1));
            if (e0 && e1) continue;
            zq0[np] = sp * nZ;
            zq1[np] = (sp + 1 < pl->p.m3) ? (sp + 1) * nZ : -1;
            if (!e0) { zin[ns] = 2 * np; eidx[sp - s0] = ns++; }
            if (!e1) { zin[ns] = 2 * np + 1; eidx[sp + 1 - s0] = ns++; }
            np++;
        }
        if (used[set]) HIPCHK(c, hipStreamWaitEvent(pl->vs, pl->evDone[set], 0)); // the wave chain has consumed this set
        if (np > 0) {
            PassArgs a1 = pass_x(pl); // P1': records -> x spectra, [pair][species] grids
            a1.out = pl->bA; a1.nspecies = nZ; a1.species_stride = pl->gsz;
            a1.recs = pl->bins.recs_sorted; a1.rowstart = pl->bins.rowstart;
            a1.q0 = zq0[0]; a1.q1 = zq1[0];
            a1.nbatch = np; a1.bstride_out = pl->gsz * (size_t)nZ;
            for (int i = 0; i < np; i++) { a1.zq0[i] = zq0[i]; a1.zq1[i] = zq1[i]; }
            HIPCHK(c, lds_pass(m1, XF_FWD, MID_ATOMS, XF_NONE, true, a1, pl->vs));
            PassArgs a2 = pass_y(pl); // P2: filter, species sum -> packed pair potentials
            a2.in0 = pl->bA; a2.gtab = pl->GT; a2.out = pl->bB; a2.nspecies = nZ; a2.species_stride = pl->gsz;
            a2.nbatch = np; a2.bstride_in0 = pl->gsz * (size_t)nZ; a2.bstride_out = pl->gsz;
            HIPCHK(c, lds_pass(m2, XF_FWD, nZ == 1 ? MID_GTAB : MID_GTABN, XF_INV, true, a2, pl->vs));
            PassArgs a3 = pass_x(pl); // P3: both transmission functions of every pair -> bCC[2 pair], bCC[2 pair + 1]
            a3.in0 = pl->bB; a3.out = pl->bCC; a3.out2 = pl->bCC + pl->gsz; a3.scale = pl->p.imPot;
            a3.band = band; a3.skip_dead_stores = bs;
            a3.nbatch = np; a3.bstride_in0 = pl->gsz; a3.bstride_out = 2 * pl->gsz; a3.bstride_out2 = 2 * pl->gsz;
            HIPCHK(c, lds_pass(m1, XF_INV, MID_EXPIV_PAIR, XF_FWD, true, a3, pl->vs));
        }
        if (ns > 0) {
            PassArgs a4 = pass_y(pl); // P4: band limit of the non-empty slices' transmission functions -> bE[set][slice]
            a4.in0 = pl->bCC; a4.out = pl->bE[set]; a4.scale = 1.f / ((float)pl->m12); a4.mindim = md;
            a4.band = band; a4.live_rows_only = bs;
            a4.nbatch = ns; a4.use_zin = 1; a4.bstride_in0 = pl->gsz; a4.bstride_out = pl->gsz;
            for (int i = 0; i < ns; i++) a4.zin[i] = zin[i];
            if (ns == 1) a4.in0 = pl->bCC + (size_t)zin[0] * pl->gsz; // a batch of one is launched without the batch offsets
            HIPCHK(c, lds_pass(m2, XF_FWD, MID_MASK, XF_INV, true, a4, pl->vs));
        }
        HIPCHK(c, hipEventRecord(pl->evReady[set], pl->vs));
        HIPCHK(c, hipStreamWaitEvent(c->stream, pl->evReady[set], 0));
        // ---- wave chain of the batch
        int s = sw > s0 ? sw : s0;
        while (s < s1) {
            if (have_seg && empty(s)) {
                int run = 1;
                RC(fused_empty_run(pl, s, nslices, &run));
                s += run;
                continue;
            }
            RC(fused_wave_step(pl, s, pl->bE[set] + (size_t)eidx[s - s0] * pl->gsz, -1));
            s++;
        }
        sw = s;
        HIPCHK(c, hipEventRecord(pl->evDone[set], c->stream));
        used[set] = true;
    }
    return FDES_OK;
}

// fork / join of the potential stream around the slices of one configuration (also inside a stream capture, where the
// event edges become graph dependencies)
int split_fork(fdes_plan* pl)
{
    if (!(pl->split && !pl->tap_mode)) return FDES_OK;
    fdes_ctx* c = pl->ctx;
    pl->p5_seen[0] = pl->p5_seen[1] = false;
    HIPCHK(c, hipEventRecord(pl->evFork, c->stream));
    HIPCHK(c, hipStreamWaitEvent(pl->vs, pl->evFork, 0));
    return FDES_OK;
}
int split_join(fdes_plan* pl)
{
    if (!(pl->split && !pl->tap_mode)) return FDES_OK;
    fdes_ctx* c = pl->ctx;
    HIPCHK(c, hipEventRecord(pl->evJoin, pl->vs));
    HIPCHK(c, hipStreamWaitEvent(c->stream, pl->evJoin, 0));
    return FDES_OK;
}

// real-space wave <-> mixed (y, kx) representation the fused loop carries between slices
int fused_enter(fdes_plan* pl)
{
    PassArgs a = pass_x(pl);
    a.in0 = pl->PSI; a.out = pl->PSIH;
    a.pitch_in = 0; a.pitch_out = pl->pitchN; // dense real-space wave -> padded mixed representation, natural store
    gang_strides(pl, a, pl->m12, pl->gsz);
    HIPCHK(pl->ctx, lds_pass(pl->p.m1, XF_FWD, MID_NONE, XF_NONE, false, a, pl->ctx->stream));
    return FDES_OK;
}
int fused_leave(fdes_plan* pl, bool propagated)
{
    PassArgs a = pass_x(pl);
    a.in0 = pl->PSIH; a.out = pl->PSI;
    if (owner_ctx(pl)->band_skip && pl->p.m1 == pl->p.m2 && propagated) { // the dead columns were last written by fused_enter: they count as zero
        const int md = pl->p.m1 < pl->p.m2 ? pl->p.m1 : pl->p.m2;
        a.band = md * md;
        a.skip_dead_loads = 1;
    }
    a.pitch_out = 0; // dense
    a.scale = 1.f / (float)pl->p.m1; // PSIH = FFT_x(psi), unnormalised transforms (m1 is a power of two: exact)
    gang_strides(pl, a, pl->gsz, pl->m12);
    HIPCHK(pl->ctx, lds_pass(pl->p.m1, XF_INV, MID_SCALE, XF_NONE, false, a, pl->ctx->stream));
    return FDES_OK;
}

// incomingWave, src/multisliceSimulation.cu:563-591
int incoming_wave(fdes_plan* pl, int k, float2* psi) // psi: where the wave goes (default: the plan's PSI)
{
    float2* const PSI = psi ? psi : pl->PSI;
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    HIPCHK(c, k_fill(PSI, pl->m12, 1.f, 0.f, c->stream));
    pl->wave_bl = !(p.mode == 2 && p.doBeamTilt);
    if (p.mode == 2) {
        HIPCHK(c, k_lens(PSI, pl->kp, p.defoci[k], c->stream));
        HIPCHK(c, fft_exec(pl,PSI, true, c->stream));
        HIPCHK(c, k_fftshift(pl->T, PSI, p.m1, p.m2, c->stream));
        HIPCHK(c, hipMemcpyAsync(PSI, pl->T, sizeof(float2) * pl->m12, hipMemcpyDeviceToDevice, c->stream));
        RC(bandwidth_limit(pl, PSI));
        HIPCHK(c, k_normalize_to(PSI, pl->m12, sqrtf((float)(p.n1 * p.n2)), pl->scal, c->stream));
    }
    if (p.doBeamTilt) HIPCHK(c, k_tilt_beam(PSI, pl->kp, p.tiltbeam[2 * k], p.tiltbeam[2 * k + 1], 1, c->stream));
    if (p.doBeamTilt && (p.mode == 0 || p.mode == 1)) {
        HIPCHK(c, k_tukey(PSI, pl->kp, c->stream));
        RC(bandwidth_limit(pl, PSI));
    }
    return FDES_OK;
}

// slice loop of one configuration up to nslices (src/crystalMaker.cu:339-344)
int slice_loop(fdes_plan* pl, int nslices)
{
    BinGeom g{pl->p.m1, pl->p.m2, pl->p.m3, pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3};
    fdes_ctx* c = pl->ctx;
    const fdes_ctx* oc = owner_ctx(pl); // lanes follow the owner's runtime options
    // the launch sequence of one configuration: fused LDS passes, or rocFFT + point-wise kernels for the other grid sizes
    auto issue = [&]() -> int {
        if (pl->fused) {
            RC(fused_enter(pl));
            RC(split_fork(pl));
            if (pl->nb > 1 && !pl->tap_mode) RC(batched_loop(pl, nslices));
            else
            for (int s = 0, adv = 1; s < nslices; s += adv) RC(fused_slice(pl, s, nslices, &adv));
            RC(split_join(pl));
            return fused_leave(pl, nslices > 0);
        }
        for (int s = 0; s < nslices; s++) {
            if ((s & 1) == 0) RC(phase_grating_pair(pl, pl->xyzFP_d, g, s));
            RC(forward_propagation(pl, s & 1));
        }
        return FDES_OK;
    };
    const bool timing_probe = (oc->probe_stride > 0);
    // the two-stream loop is issued directly: captured, its cross-stream edges cost 5 % (12.2 k against 12.85 k)
    if (!oc->opt_graph || timing_probe || nslices < 1 || (pl->fused && pl->split) || (pl->top ? pl->top : pl)->one_shot_few) return issue();
    // key: slice count, band option and the empty-slice pattern (FNV-1a over one bit per slice)
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    // the pattern itself (slice count, band option, one byte per slice) is kept beside its hash and compared on a
    // hit: a colliding hash must not replay another pattern's launch sequence
    std::vector<uint8_t> pattern;
    pattern.reserve((size_t)pl->p.m3 + 8);
    for (int b = 0; b < 4; b++) pattern.push_back((uint8_t)((unsigned)nslices >> (8 * b)));
    pattern.push_back((uint8_t)oc->band_skip);
    pattern.push_back((uint8_t)oc->walk);
    pattern.push_back((uint8_t)(oc->stagger & 255));
    pattern.push_back((uint8_t)(oc->stagger >> 8));
    pattern.push_back((uint8_t)(pl->split ? 1 : 0));
    pattern.push_back((uint8_t)(pl->wave_bl ? 1 : 0));
    pattern.push_back((uint8_t)pl->gn);
    pattern.push_back(pl->seg_h.empty() ? 0 : 1);
    if (!pl->seg_h.empty())
        for (int q = 0; q < pl->p.m3; q++) pattern.push_back(pl->seg_h[(size_t)(q + 1) * pl->nZ] == pl->seg_h[(size_t)q * pl->nZ] ? 2 : 3);
    for (uint8_t b : pattern) mix(b);
    fdes_plan::LoopGraph* gr = nullptr;
    for (auto& e : pl->graphs) if (e.key == key && e.pattern == pattern) gr = &e;
    if (!gr) {
        DeviceGuard guard(c->device);
        const int64_t skipped0 = pl->slices_skipped;
        std::vector<std::pair<int, float2*>> pow_owned;
        pl->capture_pow = &pow_owned;
        pl->capturing = true;
        hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
        int rc = FDES_OK;
        if (e == hipSuccess) rc = issue();
        hipGraph_t graph = nullptr;
        hipError_t e2 = (e == hipSuccess) ? hipStreamEndCapture(c->stream, &graph) : e;
        pl->capturing = false;
        pl->capture_pow = nullptr;
        const int64_t skipped = pl->slices_skipped - skipped0;
        pl->slices_skipped = skipped0;
        hipGraphExec_t exec = nullptr;
        hipError_t e3 = hipSuccess;
        if (rc == FDES_OK && e2 == hipSuccess) e3 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (graph) (void)hipGraphDestroy(graph);
        if (rc != FDES_OK || e2 != hipSuccess || e3 != hipSuccess) {
            for (auto& e : pow_owned) (void)hipFree(e.second);
            if (rc != FDES_OK) return rc;
            HIPCHK(c, e2);
            HIPCHK(c, e3);
        }
        if (pl->graphs.size() >= 8) { // drop the least recently used pattern
            size_t lru = 0;
            for (size_t i = 1; i < pl->graphs.size(); i++) if (pl->graphs[i].used < pl->graphs[lru].used) lru = i;
            (void)hipStreamSynchronize(c->stream); // a replay of the evicted graph may still be reading its tables
            (void)hipGraphExecDestroy(pl->graphs[lru].exec);
            for (auto& e : pl->graphs[lru].pow) (void)hipFree(e.second);
            pl->graphs.erase(pl->graphs.begin() + (long)lru);
        }
        pl->graphs.push_back({key, pattern, exec, skipped, 0, pow_owned});
        gr = &pl->graphs.back();
    }
    gr->used = ++pl->graph_tick;
    HIPCHK(c, hipGraphLaunch(gr->exec, c->stream));
    pl->slices_skipped += gr->skipped;
    return FDES_OK;
}

// exit-wave post-processing + accumulation (src/crystalMaker.cu:346-366)
int exit_wave_post(fdes_plan* pl, int k, float weight, float2* psi, float2* acc) // psi: the exit wave; acc: the intensity sum it is added to
{
    float2* const PSI = psi ? psi : pl->PSI;
    float2* const I = acc ? acc : pl->I;
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    if (pl->want_ew) HIPCHK(c, k_axpy(pl->EW, PSI, pl->m12, weight, c->stream));
    if (p.mode == 0) {
        // applyLensFunction (src/multisliceSimulation.cu:614-622) + intensityValues + Caxpy
        HIPCHK(c, fft_exec(pl,PSI, false, c->stream));
        HIPCHK(c, k_lens(PSI, pl->kp, p.defoci[k], c->stream));
        HIPCHK(c, fft_exec(pl,PSI, true, c->stream));
        HIPCHK(c, k_intensity_axpy(I, PSI, pl->m12, 1.f / ((float)pl->m12), weight, c->stream));
    } else {
        // diffractionPattern (src/crystalMaker.cu:700-718)
        if (p.doBeamTilt) HIPCHK(c, k_tilt_beam(PSI, pl->kp, p.tiltbeam[2 * k], p.tiltbeam[2 * k + 1], -1, c->stream));
        if (p.mode == 1) {
            HIPCHK(c, k_mask_filter(PSI, pl->kp, c->stream));
            RC(bandwidth_limit(pl, PSI));
        }
        HIPCHK(c, fft_exec(pl,PSI, false, c->stream));
        HIPCHK(c, k_fftshift(pl->T, PSI, p.m1, p.m2, c->stream));
        HIPCHK(c, k_intensity_axpy(I, pl->T, pl->m12, sqrtf(1.f / ((float)pl->m12)), weight, c->stream));
    }
    return FDES_OK;
}

// addNoiseAndMtf, src/crystalMaker.cu:579-613: the summed intensity in pl->I -> image k
int finalize_measurement(fdes_plan* pl, int k, float2* acc) // acc: the summed intensity (default: the plan's I)
{
    float2* const I = acc ? acc : pl->I;
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    const float alpha = 1.f / ((float)(p.m1 * p.m2));
    HIPCHK(c, fft_exec(pl,I, false, c->stream));
    if (fabsf(p.illangle) > FLT_EPSILON) {
        if (p.mode == 0) HIPCHK(c, k_spatial_incoherence(I, pl->kp, p.defoci[k], 0, c->stream));
        if (p.mode == 1 || p.mode == 2) HIPCHK(c, k_spatial_incoherence(I, pl->kp, p.defoci[k], 1, c->stream));
    }
    if (p.pD > FLT_EPSILON) {
        HIPCHK(c, k_scale(I, pl->m12, alpha, c->stream));
        HIPCHK(c, fft_exec(pl,I, true, c->stream));
        HIPCHK(c, k_noise(I, pl->m12, p.pD, (uint32_t)(1 + p.n3), k, c->stream)); // seed 1 + n3, :295
        HIPCHK(c, fft_exec(pl,I, false, c->stream));
    }
    HIPCHK(c, k_mtf(I, pl->kp, alpha, c->stream));
    HIPCHK(c, fft_exec(pl,I, true, c->stream));
    HIPCHK(c, k_crop(pl->Jout + (size_t)k * p.n1 * p.n2, I, pl->kp, c->stream));
    return FDES_OK;
}

} // namespace fdes_engine

