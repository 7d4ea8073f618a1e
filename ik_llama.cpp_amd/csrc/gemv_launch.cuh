// gemv_launch.cuh -- host-side launch geometry and template dispatch of the decode GEMV kernels (gemv.cuh).  Included only by the
// per-type instantiation TUs (gemv_inst.hip, gemv_dual.hip).
#pragma once
#include "api_internal.h"
#include "gemv.cuh"
#include <algorithm>

// ---- decode GEMV dispatch -------------------------------------------------------------------------------
// launch geometry of one GEMV: workgroups and waves per workgroup
static void gemv_grid(const cdna4_context *ctx, long M, long K, int NCOLS, int YITERS, int NR, size_t lds, unsigned grid_y, long &wgs, int &waves_per_wg, bool tables = false) {
    const int U = K >> 6; const int lpr = U <= 16 ? 16 : (U <= 32 ? 32 : 64); const int rpi = 64 / lpr;
    const long ngroups = ((long)M + rpi * NR - 1) / (rpi * NR);
    // 4 waves per workgroup; 8 when the activation vector is long enough that 256 threads would each quantize more than the
    // XPRE chunks that can be requested ahead of the weight stream (vmcnt retires in order: later chunks wait behind the weights)
    waves_per_wg = ((long)NCOLS * (K / 8) > 4L * 256) ? 8 : 4;
    static const int env_waves = getenv("CDNA4_GEMV_WAVES") ? atoi(getenv("CDNA4_GEMV_WAVES")) : 0;       // developer knobs (scripts/sweep_gemv.py)
    static const int env_per_cu = getenv("CDNA4_GEMV_PER_CU") ? atoi(getenv("CDNA4_GEMV_PER_CU")) : 0;
    if (lds > 64 * 1024) waves_per_wg = 8;              // (IQ3_S: 64 KiB bank-replicated codebook per workgroup -- at most two workgroups fit a CU, keep 16 waves on it)
    if (tables) waves_per_wg = 8;                       // codebook types: every workgroup copies its tables (12 ... 72 KiB) in the prologue -- fewer, fatter workgroups (IQ2_S 14336 x 4096: 11.4 -> 10.8 us, fused 17.0 -> 15.6 us, scripts/r03_gpu20.sh)
    if (env_waves) waves_per_wg = env_waves;
    // Workgroup count: a multiple of the CU count (every CU gets the same number of workgroups) chosen so that the row
    // groups divide as evenly as possible over the waves (a wave with one extra row group is pure tail), preferring
    // fewer workgroups (each pays the activation-quantize prologue) as long as a wave keeps <= ~8 row groups.
    const long lds_fit = std::max<long>(1, (long)(160 * 1024 / std::max<size_t>(lds, 1)));
    const long max_per_cu = std::min<long>(lds_fit, (lds > 40 * 1024 || YITERS >= 4 || waves_per_wg == 8 || NR > 1 || (NCOLS > 1 && YITERS > 0)) ? 2 : 4);   // (register-heavy variants: <= 2-3 waves / SIMD)      // (NR = 2 kernels hold > 128 VGPRs: <= 3 waves / SIMD)
    if (grid_y > 1) { wgs = std::max<long>(1, std::min<long>((ngroups + waves_per_wg - 1) / waves_per_wg, (ctx->num_cu * max_per_cu + grid_y - 1) / grid_y)); }
    else if (ngroups <= (long)ctx->num_cu * waves_per_wg) wgs = (ngroups + waves_per_wg - 1) / waves_per_wg;
    else {
        long best = 1; double best_cost = 1e30;
        for (long per_cu = 1; per_cu <= max_per_cu; ++per_cu) {
            const long waves = per_cu * ctx->num_cu * waves_per_wg;
            const long rpw = (ngroups + waves - 1) / waves;                   // row groups of the busiest wave
            const double cost = (double)rpw * waves / (double)ngroups + 0.04 * per_cu + (rpw > 8 ? 0.02 * (rpw - 8) : 0.0);
            if (cost < best_cost) { best_cost = cost; best = per_cu; }
        }
        if (env_per_cu) best = env_per_cu;
        wgs = best * ctx->num_cu;
    }
}
#ifdef GEMV_EXP_TIMELINE      // experiment builds only (scripts/gemv_timeline.py): per-workgroup phase stamps of the next GEMV launches
extern long long *g_gemv_timeline; extern int g_gemv_timeline_wgs;       // (cdna4_api.hip)
#endif
template <int TYPE, int NCOLS, bool UPGATE, int YITERS, int VDT, int DEPTH, bool MULTI, int NR, int LPR>
static int launch_gemv_lpr(const GemvArgs &a, long wgs, unsigned grid_y, int waves_per_wg, size_t lds, hipStream_t st) {
    cdna4_note_gemv("gemv", TYPE, NCOLS, (int)UPGATE, YITERS, NR, LPR, a.norm_w ? (a.rope_tab ? 4 : 1) : (a.R ? 2 : 0), wgs, grid_y, waves_per_wg);      // (tests assert the geometry: cdna4_last_launch_info)
    // graph-level fusions of a decoded token (separate instantiations: the plain kernels stay byte-identical): FX = 1 RMS norm of the activation row in the
    // prologue (needs the whole row in the pre-loaded chunks), FX = 2 residual add in the epilogue
    if constexpr (NCOLS == 1 && YITERS == 1) {
        if (a.norm_w) {
            if (a.R || a.q8_out || !a.src_f32 || a.ids || (long)(a.K >> 3) > (long)xpre_for(NCOLS, YITERS, LPR, type_has_tables(type_base(TYPE))) * 64 * waves_per_wg) return set_err(CDNA4_E_UNSUPPORTED, "gemv: fused norm needs one f32 row of <= %d values", 8 * xpre_for(NCOLS, YITERS, LPR, type_has_tables(type_base(TYPE))) * 64 * waves_per_wg);
            const size_t ldn = lds + 64;
            if (a.rope_tab) {        // + the q,k,v epilogue (FX = 4): ROPE of the Q / K rows, K / V rows to the f16 cache
                if constexpr (MULTI && NR == 1 && LPR == 64 && !UPGATE) {
                    if (a.M % 2) return set_err(CDNA4_E_UNSUPPORTED, "gemv: q,k,v epilogue needs an even row count");
                    if (ldn > 64 * 1024) { const int rc = cdna4_opt_in_lds((const void *)gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR, 4>); if (rc) return rc; }
                    hipLaunchKernelGGL((gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR, 4>), dim3((unsigned)wgs, grid_y), dim3(64 * waves_per_wg), ldn, st, a);
                    HIP_TRY(hipGetLastError()); return CDNA4_OK;
                } else return set_err(CDNA4_E_UNSUPPORTED, "gemv: no q,k,v epilogue variant for this launch shape");
            }
            if (ldn > 64 * 1024) { const int rc = cdna4_opt_in_lds((const void *)gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR, 1>); if (rc) return rc; }
            hipLaunchKernelGGL((gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR, 1>), dim3((unsigned)wgs, grid_y), dim3(64 * waves_per_wg), ldn, st, a);
            HIP_TRY(hipGetLastError()); return CDNA4_OK;
        }
        if constexpr (!UPGATE && !MULTI) {
            if (a.R) {
                if (a.q8_out || a.ids) return set_err(CDNA4_E_UNSUPPORTED, "gemv: fused residual on a plain mat-mul only");
                if (lds > 64 * 1024) { const int rc = cdna4_opt_in_lds((const void *)gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR, 2>); if (rc) return rc; }
                hipLaunchKernelGGL((gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR, 2>), dim3((unsigned)wgs, grid_y), dim3(64 * waves_per_wg), lds, st, a);
                HIP_TRY(hipGetLastError()); return CDNA4_OK;
            }
        }
    }
    if (a.norm_w || a.R || a.rope_tab) return set_err(CDNA4_E_UNSUPPORTED, "gemv: no fused norm / residual variant for this launch shape");
    if (lds > 64 * 1024) { const int rc = cdna4_opt_in_lds((const void *)gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR>); if (rc) return rc; }
    hipLaunchKernelGGL((gemv_kernel<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, LPR>), dim3((unsigned)wgs, grid_y), dim3(64 * waves_per_wg), lds, st, a);
    HIP_TRY(hipGetLastError());
    return CDNA4_OK;
}
template <int TYPE, int NCOLS, bool UPGATE, int YITERS, int VDT, int DEPTH = GEMV_DEPTH, bool MULTI = false, int NR = 1>
static int launch_gemv_y(const cdna4_context *ctx, const GemvArgs &a, unsigned grid_y, hipStream_t st) {
    const bool emit = UPGATE && NR == 2 && NCOLS == 1 && a.q8_out != nullptr;
    if (a.q8_out && !emit) return set_err(CDNA4_E_UNSUPPORTED, "quantized result emission is only available on the fused two-row decode kernel");
    const size_t lds = gemv_lds_bytes<VDT>(NCOLS, a.K, type_base(TYPE)) + (emit ? 256 : 0);
    long wgs; int waves_per_wg;
    gemv_grid(ctx, a.M, a.K, NCOLS, YITERS, NR, lds, grid_y, wgs, waves_per_wg, type_has_tables(type_base(TYPE)));
    if (emit) { wgs = ((long)a.M + 63) / 64; waves_per_wg = 8; }          // one workgroup per 64 consecutive rows (two q8 blocks)
    // (fused two-row launch, 14336 rows = 7168 pairs: with 512 x 4 waves a wave walks 3.5 pairs on average -- balanced geometries were measured in round 5 and are NOT faster:
    //  256 x 7 waves 15.71 us, 256 x 14 15.48, 256 x 8 16.48, 512 x 7 19.0 against 15.39 us for the default, profiles/r05_notes.md)
#ifdef GEMV_EXP_TIMELINE
    const_cast<GemvArgs &>(a).timeline = g_gemv_timeline; g_gemv_timeline_wgs = (int)wgs;
#endif
    // single-column launches on rows of more than 32 units (K > 2048): 64 lanes per row known at compile time
    if constexpr (NCOLS == 1) { if ((a.K >> 6) > 32) return launch_gemv_lpr<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, 64>(a, wgs, grid_y, waves_per_wg, lds, st); }
    return launch_gemv_lpr<TYPE, NCOLS, UPGATE, YITERS, VDT, DEPTH, MULTI, NR, 0>(a, wgs, grid_y, waves_per_wg, lds, st);
}
template <int TYPE, bool UPGATE, int VDT>
static int launch_gemv_t(const cdna4_context *ctx, const GemvArgs &a, int ncols, unsigned grid_y, hipStream_t st) {
    if (ncols == 1) {      // single column: activations live in registers when a row is <= 4 slices of 64 lanes
        const int U = a.K >> 6, iters = U <= 64 ? 1 : (U + 63) / 64;
        // two rows per step (shared activations + bookkeeping) once there are enough row groups to give every wave of a full grid work
        static const int env_nr = getenv("CDNA4_GEMV_NR") ? atoi(getenv("CDNA4_GEMV_NR")) : 0;
        const int lpr = U <= 16 ? 16 : (U <= 32 ? 32 : 64);
        // measured (profiles/r01_notes.md): pays on the fused up*gate launch and on very tall matrices (output.weight, >= 24 rows per wave
        // of a full grid); on 4096..14336-row matrices the halved wave count costs more latency hiding than the instructions saved
        // (codebook types, fused launch: one row pair per step is SLOWER -- their step is the chain gather addresses -> 24-32 LDS gathers -> sign / dot, and two rows x (up, gate)
        //  put four such chains back to back in one wave where single rows leave them to different waves: IQ2_S 20.1 -> 17.5 us, IQ3_S 21.7 -> 20.7 us at 2 x 14336 x 4096,
        //  scripts/iq_exp.py; the q8-emitting form needs the pair and keeps it)
        const bool nr2 = env_nr ? env_nr == 2 : (UPGATE ? (!type_has_tables(TYPE) || a.q8_out) && (long)a.M * lpr / 64 >= 2L * 4 * ctx->num_cu * 2 : (long)a.M * lpr / 64 >= 24L * 8 * ctx->num_cu);
        if constexpr (!UPGATE) {
            if (a.nmat > 1) {       // fused q,k,v launch: per-row matrix lookup compiled in only here
                if (iters == 1) return nr2 ? launch_gemv_y<TYPE, 1, false, 1, VDT, GEMV_DEPTH, true, 2>(ctx, a, grid_y, st) : launch_gemv_y<TYPE, 1, false, 1, VDT, GEMV_DEPTH, true>(ctx, a, grid_y, st);
                if (iters == 2) return launch_gemv_y<TYPE, 1, false, 2, VDT, GEMV_DEPTH, true>(ctx, a, grid_y, st);
                if (iters <= 4) return launch_gemv_y<TYPE, 1, false, 4, VDT, GEMV_DEPTH, true>(ctx, a, grid_y, st);
                return launch_gemv_y<TYPE, 1, false, 0, VDT, GEMV_DEPTH, true>(ctx, a, grid_y, st);
            }
            if (nr2) {
                if (iters == 1) return launch_gemv_y<TYPE, 1, false, 1, VDT, GEMV_DEPTH, false, 2>(ctx, a, grid_y, st);
                if (iters == 2) return launch_gemv_y<TYPE, 1, false, 2, VDT, GEMV_DEPTH, false, 2>(ctx, a, grid_y, st);
                if (iters <= 4 && TYPE != T_Q5_K) return launch_gemv_y<TYPE, 1, false, 4, VDT, GEMV_DEPTH, false, 2>(ctx, a, grid_y, st);     // (Q5_K: would spill)
            }
        } else {
            if ((nr2 || a.q8_out) && iters == 1) return launch_gemv_y<TYPE, 1, true, 1, VDT, 2, false, 2>(ctx, a, grid_y, st);       // up+gate x 2 rows: ring of 2 keeps 8 units in flight
            if constexpr (type_has_tables(TYPE)) {      // codebook types: single rows (see nr2) with a ring of 2 (up + gate = 4 units in flight per lane; 145 instead of 173 registers)
                constexpr int env_d2 = 1;      // (round 3 A/B closed: IQ2_S 17.0 -> 16.4, IQ3_S 20.7 -> 18.8 us)
                if (env_d2 && iters == 1) return launch_gemv_y<TYPE, 1, true, 1, VDT, 2, false, 1>(ctx, a, grid_y, st);
            }
        }
        if constexpr (!UPGATE && !type_has_tables(TYPE)) {
            // long rows (ffn_down, K = 2..4 slices of 4096): two rows per wave walked slice-major, activations quantized slice by slice
            static const int env_sliced = getenv("CDNA4_GEMV_SLICED") ? atoi(getenv("CDNA4_GEMV_SLICED")) : 1;
            const long wgs = a.M / 16;
            // (measured, profiles/r01_notes.md: Q4_K 10.3 -> 9.7 us, Q6_K 14.8 -> 13.8 us at 4096 x 14336; the codebook types lose 3 %:
            //  their per-step LDS gathers, not the prologue, are what the waves wait on)
            if (env_sliced && iters >= 2 && iters <= 4 && a.src_f32 && !a.ids && !a.norm_w && grid_y == 1 && a.M % 16 == 0 && 2 * wgs >= ctx->num_cu && wgs <= 2L * ctx->num_cu) {
                const size_t lds = gemv_lds_bytes<VDT>(1, a.K, type_base(TYPE));
                cdna4_note_gemv("gemv_sliced", TYPE, 1, 0, iters, 2, 64, a.R ? 2 : 0, wgs, 1, 8);
                if (a.R) hipLaunchKernelGGL((gemv_sliced_kernel<TYPE, VDT, 8, 4, 2>), dim3((unsigned)wgs), dim3(512), lds, st, a);
                else     hipLaunchKernelGGL((gemv_sliced_kernel<TYPE, VDT, 8, 4>), dim3((unsigned)wgs), dim3(512), lds, st, a);
                HIP_TRY(hipGetLastError());
                return CDNA4_OK;
            }
        }
        if (iters == 1) return launch_gemv_y<TYPE, 1, UPGATE, 1, VDT>(ctx, a, grid_y, st);
        if (iters == 2) return launch_gemv_y<TYPE, 1, UPGATE, 2, VDT>(ctx, a, grid_y, st);
        // (a ring of 8 units for long rows -- a wave's whole share requested up front, activations from LDS -- measured 1.5-2 us
        //  SLOWER than ring 4 + register-resident activations on the K = 14336 down projections: profiles/r01_notes.md)
        if (iters <= 4) return launch_gemv_y<TYPE, 1, UPGATE, 4, VDT>(ctx, a, grid_y, st);
        return launch_gemv_y<TYPE, 1, UPGATE, 0, VDT>(ctx, a, grid_y, st);
    }
    // 2..4 columns of a single K-slice (K <= 4096): the lane's activation slices of all columns stay in registers (80 VGPRs at 4 columns)
    constexpr int env_mcreg = 2;      // register-resident activations for plain AND fused 2-4 column launches (round 2 A/B closed)
    // (measured, 14336 x 4096: Q4_K N = 2 / 4 13.3 -> 11.9 / 17.7 -> 15.0 us, 8 columns 33.8 -> 29.2; fused N = 2 / 4 21.7 -> 20.7 / 27.5 -> 26.2 us)
    if ((UPGATE ? env_mcreg >= 2 : env_mcreg >= 1) && (a.K >> 6) <= 64 && a.nmat <= 1) {
        switch (ncols) {
            case 2: return launch_gemv_y<TYPE, 2, UPGATE, 1, VDT>(ctx, a, grid_y, st);
            case 3: return launch_gemv_y<TYPE, 3, UPGATE, 1, VDT>(ctx, a, grid_y, st);
            case 4: return launch_gemv_y<TYPE, 4, UPGATE, 1, VDT>(ctx, a, grid_y, st);
        }
    }
    switch (ncols) {
        case 2: return launch_gemv_y<TYPE, 2, UPGATE, 0, VDT>(ctx, a, grid_y, st);
        case 3: return launch_gemv_y<TYPE, 3, UPGATE, 0, VDT>(ctx, a, grid_y, st);
        case 4: return launch_gemv_y<TYPE, 4, UPGATE, 0, VDT>(ctx, a, grid_y, st);
    }
    return set_err(CDNA4_E_INVALID, "gemv: ncols %d", ncols);
}
