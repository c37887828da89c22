// gangs.hip — gangs (the configurations of a measurement, or the measurements of a series, in lockstep: every pass ONE launch
// with the members as grid z), lanes (configurations in flight on streams of their own) and the progress report.
// Split from engine.hip in round 5; shared declarations: engine_impl.h.
#include "engine_impl.h"

namespace fdes_engine {

// 2-D transforms of n grids `stride` elements apart: with the hand-written back-end the two passes take the grids as grid z
int fft_gang(fdes_plan* pl, float2* data, int n, size_t stride, bool inverse)
{
    fdes_ctx* c = pl->ctx;
    if (n <= 1 || pl->fft->backend != 2 || !pl->gscr) {
        for (int g = 0; g < n; g++) HIPCHK(c, fft_exec(pl, data + (size_t)g * stride, inverse, c->stream));
        return FDES_OK;
    }
    const int xf = inverse ? XF_INV : XF_FWD;
    PassArgs a;
    a.in0 = data; a.out = pl->gscr; a.tw0 = pl->fft->tw0x; a.tw1 = pl->fft->tw1x; a.jit = pl->fft->jit_x; a.tile_rows = pl->fft->rows_x; a.nrows = pl->p.m2; a.wg = pl->fft->wg;
    a.nbatch = n; a.bstride_in0 = stride; a.bstride_out = pl->m12;
    HIPCHK(c, lds_pass(pl->p.m1, xf, MID_NONE, XF_NONE, true, a, c->stream));
    PassArgs b;
    b.in0 = pl->gscr; b.out = data; b.tw0 = pl->fft->tw0y; b.tw1 = pl->fft->tw1y; b.jit = pl->fft->jit_y; b.tile_rows = pl->fft->rows_y; b.nrows = pl->p.m1; b.wg = pl->fft->wg;
    b.nbatch = n; b.bstride_in0 = pl->m12; b.bstride_out = stride;
    HIPCHK(c, lds_pass(pl->p.m2, xf, MID_NONE, XF_NONE, true, b, c->stream));
    return FDES_OK;
}

// bandwidthLimit of n grids m12 apart
int bandwidth_limit_gang(fdes_plan* pl, float2* f, int n)
{
    RC(fft_gang(pl, f, n, pl->m12, false));
    HIPCHK(pl->ctx, k_mask_scale_gang(f, pl->m12, n, pl->p.m1, pl->p.m2, 1.f / ((float)pl->m12), pl->ctx->stream));
    return fft_gang(pl, f, n, pl->m12, true);
}

// incoming_wave of the n members (PSI, m12 apart).  Members of one k share their wave: built once, copied.  Members with a k
// of their own (gangs across measurements): plane waves with a beam tilt take one launch per step for all members; a
// CBED probe (own lens, own norm) is built member by member.
int incoming_wave_gang(fdes_plan* pl, int n)
{
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    float2* const psi0 = pl->PSI;
    bool same_k = true;
    for (int g = 1; g < n; g++) same_k = same_k && pl->gq[(size_t)g].k == pl->gq[0].k;
    if (n > 1 && !same_k && p.mode != 2 && pl->gscr) {
        pl->wave_bl = true;
        HIPCHK(c, k_fill(psi0, pl->m12 * (size_t)n, 1.f, 0.f, c->stream));
        if (p.doBeamTilt) {
            GangPar tb0, tb1;
            tb0.n = tb1.n = n;
            for (int g = 0; g < n; g++) { tb0.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k]; tb1.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k + 1]; }
            HIPCHK(c, k_tilt_beam_gang(psi0, pl->m12, pl->kp, tb0, tb1, 1, c->stream));
            HIPCHK(c, k_tukey_gang(psi0, pl->m12, n, pl->kp, c->stream));
            RC(bandwidth_limit_gang(pl, psi0, n));
        }
        return FDES_OK;
    }
    int rcw = FDES_OK;
    for (int g = 0; g < n && rcw == FDES_OK; g++) {
        float2* const mine = psi0 + (size_t)g * pl->m12;
        if (g > 0 && pl->gq[(size_t)g].k == pl->gq[(size_t)g - 1].k) {
            if (hipMemcpyAsync(mine, mine - pl->m12, sizeof(float2) * pl->m12, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) rcw = FDES_EGPU;
        } else {
            rcw = incoming_wave(pl, pl->gq[(size_t)g].k, mine);
        }
    }
    return rcw;
}

// exit_wave_post of the n members of a gang (waves back to back in PSI) in a handful of launches (no exit-wave output:
// that case goes member by member)
int exit_wave_post_gang(fdes_plan* pl, int n)
{
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    if (n <= 1 || pl->want_ew || !pl->gscr) {
        int rce = FDES_OK;
        for (int g = 0; g < n && rce == FDES_OK; g++)
            rce = exit_wave_post(pl, pl->gq[(size_t)g].k, pl->gq[(size_t)g].w, pl->PSI + (size_t)g * pl->m12, pl->I + (size_t)pl->gq[(size_t)g].slot * pl->m12);
        return rce;
    }
    GangPar dk, wt;
    dk.n = wt.n = n;
    for (int g = 0; g < n; g++) {
        dk.f[g] = p.defoci[pl->gq[(size_t)g].k];
        wt.f[g] = pl->gq[(size_t)g].w;
        wt.k[g] = pl->gq[(size_t)g].slot;
    }
    if (p.mode == 0) {
        RC(fft_gang(pl, pl->PSI, n, pl->m12, false));
        HIPCHK(c, k_lens_gang(pl->PSI, pl->m12, pl->kp, dk, c->stream));
        RC(fft_gang(pl, pl->PSI, n, pl->m12, true));
        HIPCHK(c, k_intensity_gang(pl->I, pl->PSI, pl->m12, 1.f / ((float)pl->m12), wt, c->stream));
        return FDES_OK;
    }
    // diffractionPattern (src/crystalMaker.cu:700-718), as exit_wave_post
    if (p.doBeamTilt) {
        GangPar tb0, tb1;
        tb0.n = tb1.n = n;
        for (int g = 0; g < n; g++) { tb0.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k]; tb1.f[g] = p.tiltbeam[2 * pl->gq[(size_t)g].k + 1]; }
        HIPCHK(c, k_tilt_beam_gang(pl->PSI, pl->m12, pl->kp, tb0, tb1, -1, c->stream));
    }
    if (p.mode == 1) {
        HIPCHK(c, k_mask_filter_gang(pl->PSI, pl->m12, n, pl->kp, c->stream));
        RC(bandwidth_limit_gang(pl, pl->PSI, n));
    }
    RC(fft_gang(pl, pl->PSI, n, pl->m12, false));
    HIPCHK(c, k_fftshift_gang(pl->gscr, pl->PSI, pl->m12, n, p.m1, p.m2, c->stream)); // (the transforms are done with their scratch)
    HIPCHK(c, k_intensity_gang(pl->I, pl->gscr, pl->m12, sqrtf(1.f / ((float)pl->m12)), wt, c->stream));
    return FDES_OK;
}

// finalize_measurement of the measurements in gfinal when their slots are 0, 1, 2 ... in order (what
// fdes_build_measurements queues): the detector chain over all slots per launch
int finalize_gang(fdes_plan* pl)
{
    fdes_ctx* c = pl->ctx;
    const fdes_params& p = pl->p;
    const int n = (int)pl->gfinal.size();
    bool in_order = n > 1 && n <= 16 && pl->gscr != nullptr;
    for (int q = 0; q < n && in_order; q++) in_order = pl->gfinal[(size_t)q].second == q;
    if (!in_order) {
        int rcf = FDES_OK;
        for (size_t q = 0; q < pl->gfinal.size() && rcf == FDES_OK; q++)
            rcf = finalize_measurement(pl, pl->gfinal[q].first, pl->I + (size_t)pl->gfinal[q].second * pl->m12);
        return rcf;
    }
    GangPar dk, kk;
    dk.n = kk.n = n;
    for (int q = 0; q < n; q++) { dk.f[q] = p.defoci[pl->gfinal[(size_t)q].first]; kk.k[q] = pl->gfinal[(size_t)q].first; }
    const float alpha = 1.f / ((float)(p.m1 * p.m2));
    RC(fft_gang(pl, pl->I, n, pl->m12, false));
    if (fabsf(p.illangle) > FLT_EPSILON) HIPCHK(c, k_spatial_incoherence_gang(pl->I, pl->m12, pl->kp, p.mode == 0 ? 0 : 1, dk, c->stream));
    if (p.pD > FLT_EPSILON) {
        HIPCHK(c, k_scale(pl->I, pl->m12 * (size_t)n, alpha, c->stream));
        RC(fft_gang(pl, pl->I, n, pl->m12, true));
        HIPCHK(c, k_noise_gang(pl->I, pl->m12, pl->m12, p.pD, (uint32_t)(1 + p.n3), kk, c->stream));
        RC(fft_gang(pl, pl->I, n, pl->m12, false));
    }
    HIPCHK(c, k_mtf_gang(pl->I, pl->m12, n, pl->kp, alpha, c->stream));
    RC(fft_gang(pl, pl->I, n, pl->m12, true));
    HIPCHK(c, k_crop_gang(pl->Jout, pl->I, pl->m12, pl->kp, kk, c->stream));
    return FDES_OK;
}

// The queued configurations of this plan as ONE gang: the incoming wave once per measurement k (it depends on k only;
// members of the same k get a copy), atoms / records per member, one slice loop with the members as grid z, the detector
// chain per member into the member's intensity slot.  The members are the configurations of one measurement - or, for a
// series with one configuration per measurement (gang_k), measurements.  A slice counts as empty (skip_empty) only when
// it is empty in every member: the others run the full sequence on it, which is always correct (t = band-limited 1).
int gang_flush(fdes_plan* pl)
{
    const int n = (int)pl->gq.size();
    fdes_ctx* c = pl->ctx;
    if (n > 0) {
        RC(incoming_wave_gang(pl, n));
        if (pl->nAt > 0) {
            // tilt, jitter and binning of all members in one launch each (geometry.hip, *_gang): what config_atoms does
            // member by member, to the bit
            int ks[16], js[16];
            float t0[16], t1[16];
            bool same_k = true;
            for (int g = 0; g < n; g++) {
                ks[g] = pl->gq[(size_t)g].k; js[g] = pl->gq[(size_t)g].j;
                t0[g] = pl->p.tiltspec[2 * ks[g]]; t1[g] = pl->p.tiltspec[2 * ks[g] + 1];
                same_k = same_k && ks[g] == ks[0];
            }
            const size_t n3f = 3 * (size_t)pl->nAt;
            if (same_k) {
                RC(ensure_tilt(pl, ks[0]));
                if (pl->p.frPh > 0) HIPCHK(c, geom_jitter_gang(pl->gxyzFP, pl->xyzK_d, 0, pl->dwf_d, pl->nAt, n, owner_ctx(pl)->seed, ks, js, c->stream));
                else for (int g = 0; g < n; g++) HIPCHK(c, hipMemcpyAsync(pl->gxyzFP + (size_t)g * n3f, pl->xyzK_d, sizeof(float) * n3f, hipMemcpyDeviceToDevice, c->stream));
            } else {
                HIPCHK(c, geom_tilt_gang(pl->gxyzFP, pl->xyzTO_d, pl->nAt, n, t0, t1, c->stream));
                if (pl->p.frPh > 0) HIPCHK(c, geom_jitter_gang(pl->gxyzFP, pl->gxyzFP, n3f, pl->dwf_d, pl->nAt, n, owner_ctx(pl)->seed, ks, js, c->stream));
            }
            BinGeom bg{pl->p.m1, pl->p.m2, pl->p.m3, pl->nZ, pl->p.d1, pl->p.d2, pl->p.d3};
            HIPCHK(c, geom_bin_atoms_gang(pl->gxyzFP, pl->spec_d, pl->occ_d, pl->nAt, n, bg, pl->bins, pl->seg_stride, pl->rowstart_stride, c->stream));
        } else
        for (int g = 0; g < n; g++)
            RC(config_atoms(pl, pl->gq[(size_t)g].k, pl->gq[(size_t)g].j, false, pl->gxyzFP + (size_t)g * 3 * (size_t)pl->nAt, &pl->gbins[(size_t)g]));
        // which slices hold atoms: one question (n small copies, ONE host wait) for the whole gang; the rules of
        // config_atoms for when a dense specimen is no longer asked, counted per member
        bool have_all = false;
        {
            fdes_plan* tp = pl->top ? pl->top : pl;
            constexpr int kDenseAfter = 8, kDenseRecheck = 64;
            bool ask = owner_ctx(pl)->skip_empty != 0;
            if (ask && tp->dense_streak >= kDenseAfter && (tp->cfg_seen % kDenseRecheck) >= n) ask = false;
            tp->cfg_seen += n;
            if (ask) {
                std::vector<int> qk((size_t)n), qj((size_t)n);
                for (int g = 0; g < n; g++) { qk[(size_t)g] = pl->gq[(size_t)g].k; qj[(size_t)g] = pl->gq[(size_t)g].j; }
                RC(empty_query(pl, n, qk.data(), qj.data())); // fills seg_h (occupied in ANY member) and, for n > 1, gseg[g]
                tp->empty_queries++;
                for (int g = 0; g < n; g++) {
                    const std::vector<int>& t = n > 1 ? pl->gseg[(size_t)g] : pl->seg_h;
                    bool any_empty = false;
                    for (int q = 0; q < pl->p.m3 && !any_empty; q++) any_empty = t[(size_t)(q + 1) * pl->nZ] == t[(size_t)q * pl->nZ];
                    tp->dense_streak = any_empty ? 0 : tp->dense_streak + 1;
                }
                have_all = true;
            }
        }
        if (!have_all) pl->seg_h.clear();
        if (pl->ev_used == pl->evs.size()) {
            EvPair e{};
            HIPCHK(c, hipEventCreate(&e.a));
            HIPCHK(c, hipEventCreate(&e.b));
            pl->evs.push_back(e);
        }
        EvPair& ev = pl->evs[pl->ev_used++];
        ev.slices = pl->p.m3 * n;
        ev.configs = n;
        HIPCHK(c, hipEventRecord(ev.a, c->stream));
        pl->gn = n;
        const int rcl = slice_loop(pl, pl->p.m3);
        pl->gn = 1;
        RC(rcl);
        HIPCHK(c, hipEventRecord(ev.b, c->stream));
        pl->slices_done += (int64_t)pl->p.m3 * n;
        const int rce = exit_wave_post_gang(pl, n);
        pl->gq.clear();
        RC(rce);
    }
    // measurements whose last member has just been issued: detector chain on their slot
    if (!pl->gfinal.empty()) {
        const int rcf = finalize_gang(pl);
        pl->gfinal.clear();
        RC(rcf);
    }
    return FDES_OK;
}

// everything queued on this plan and its lanes is issued (before anything reads or resets the sums)
int gang_flush_all(fdes_plan* pl)
{
    RC(gang_flush(pl));
    for (fdes_plan* l : pl->lanes) {
        const int rc = gang_flush(l);
        if (rc != FDES_OK) { pl->ctx->err = "lane: " + l->ctx->err; return rc; }
    }
    return FDES_OK;
}

// Lane 0 takes over the partial sums of the other lanes: I += I_lane (and the exit-wave sum), ordered by
// events in both directions (lane stream -> lane 0 before the read, lane 0 -> lane stream before the lane
// reuses its accumulators).
int fold_lanes(fdes_plan* pl)
{
    fdes_ctx* c = pl->ctx;
    RC(gang_flush_all(pl));
    if (!pl->lanes_dirty) return FDES_OK;
    for (size_t l = 0; l < pl->lanes.size(); l++) {
        fdes_plan* lp = pl->lanes[l];
        HIPCHK(c, hipEventRecord(pl->lane_ev[l], lp->ctx->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, pl->lane_ev[l], 0));
        HIPCHK(c, k_axpy(pl->I, lp->I, pl->m12, 1.f, c->stream));
        HIPCHK(c, k_fill(lp->I, pl->m12, 0.f, 0.f, c->stream));
        if (pl->want_ew) {
            HIPCHK(c, k_axpy(pl->EW, lp->EW, pl->m12, 1.f, c->stream));
            HIPCHK(c, k_fill(lp->EW, pl->m12, 0.f, 0.f, c->stream));
        }
        HIPCHK(c, hipEventRecord(pl->lane_ev[l], c->stream));
        HIPCHK(c, hipStreamWaitEvent(lp->ctx->stream, pl->lane_ev[l], 0));
    }
    pl->lanes_dirty = false;
    return FDES_OK;
}

// configurations whose slice loop has finished on the GPU (this plan and its lanes); oldest_pending = the end event of
// the oldest one still running, if any
int64_t configs_finished(fdes_plan* pl, hipEvent_t* oldest_pending)
{
    int64_t n = 0;
    auto scan = [&](fdes_plan* q) {
        while (q->ev_done < q->ev_used && hipEventQuery(q->evs[q->ev_done].b) == hipSuccess) q->cfg_done += q->evs[q->ev_done++].configs;
        (void)hipGetLastError(); // hipErrorNotReady is not an error
        n += q->cfg_done;
        if (oldest_pending && !*oldest_pending && q->ev_done < q->ev_used) *oldest_pending = q->evs[q->ev_done].b;
    };
    scan(pl);
    for (fdes_plan* l : pl->lanes) scan(l);
    return n;
}

// rate-limited progress report of fdes_build_measurements; also bounds the number of configurations in flight
void report_progress(fdes_plan* pl, int64_t issued, int64_t total_configs, bool final)
{
    fdes_ctx* c = pl->ctx;
    if (!c->progress) return;
    const int64_t depth = 2 * (int64_t)(pl->lanes.size() + 1);
    int64_t done = 0;
    for (;;) {
        hipEvent_t pending = nullptr;
        done = configs_finished(pl, &pending);
        if ((!final && issued - done <= depth) || !pending) break;
        (void)hipEventSynchronize(pending);
    }
    const auto now = std::chrono::steady_clock::now();
    if (!final && std::chrono::duration_cast<std::chrono::milliseconds>(now - c->progress_last).count() < c->progress_min_ms) return;
    c->progress_last = now;
    c->progress(c->progress_user, done * (int64_t)pl->p.m3, total_configs * (int64_t)pl->p.m3);
}

} // namespace fdes_engine

