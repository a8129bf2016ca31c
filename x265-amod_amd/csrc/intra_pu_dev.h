/* The fused prediction-unit step of the intra RD (include/x265amd.h: x265amd_intra_pu): shared by its kernel (tu_kernels.hip) and the job server
 * (device_queue.hip). */
#ifndef X265AMD_INTRA_PU_DEV_H
#define X265AMD_INTRA_PU_DEV_H
#include "tu_dev.h"
#include "intra_dev.h"

/* x265amd_intra_pu: scan, candidate list, candidate chains by one workgroup (see include/x265amd.h).  smem: the larger of IntraScanLds and one
 * TuLds + IntraTuLds per wavefront. */
XA_DEV void block_intra_pu(const x265amd_intra_pu_job* pj, x265amd_intra_pu_out* po, x265amd_tu_result* res, char* smem, int tid, int nthr)
{
    __shared__ int32_t s_sa8d[35];
    __shared__ uint8_t s_modes[16];
    __shared__ int s_num;
    XA_STAGE(15);
    const x265amd_intra_pu_job P = xa_ld_record(pj);
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    XA_STAGE(13);
    {
        x265amd_intra_job sj;
        sj.recon = P.tmpl.nb; sj.fenc = P.tmpl.tu.fenc; sj.avail = P.tmpl.avail; sj.recon_stride = P.tmpl.nb_stride; sj.fenc_stride = P.tmpl.tu.fenc_stride;
        sj.log2_tr_size = P.tmpl.tu.log2_tr_size; sj.strong_smoothing = P.tmpl.strong_smoothing;
        block_intra_scan_job(sj, s_sa8d, nullptr, *reinterpret_cast<IntraScanLds*>(smem), tid, nthr);
    }
    __syncthreads();
    XA_STAGE(14);
    if (tid < 35) po->sa8d[tid] = s_sa8d[tid];
    /* estIntraPredQT (search.cpp:1615-1650): costs (a lane per mode), the padded best, then updateCandList over the eligible modes in mode order.  Costs fit
     * 32 bits with room to spare (sa8d of a 32x32 block of 12-bit samples < 2^28, bits x lambda >> 8 < 2^14), so the comparisons are the reference's. */
    if (tid < 64)
    {
        const uint32_t kMax = 0xFFFFFFFFu;
        uint32_t myCost = kMax;
        if (tid < 35)
        {
            uint32_t b = P.rbits;
            if (tid == P.preds[0]) b = P.mpm_base + 1u;
            else if (tid == P.preds[1] || tid == P.preds[2]) b = P.mpm_base + 2u;
            myCost = (uint32_t)s_sa8d[tid] + (uint32_t)(((unsigned long long)b * P.lambda + 128) >> 8);
        }
        uint32_t bcost = myCost;
        for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(bcost, off, 64); bcost = o < bcost ? o : bcost; }
        const uint32_t padded = bcost + (bcost >> 2);
        unsigned long long todo = __ballot(tid < 35 && (myCost < padded || tid == P.preds[0]));
        const int maxCand = P.max_cand > 16 ? 16 : P.max_cand;
        const int numEligible = __popcll(todo);
        /* the first maxCand eligible modes take the places in order (each replaces the first empty place) */
        const int myRank = __popcll(todo & ((1ull << tid) - 1));
        const bool mineEligible = tid < 35 && ((todo >> tid) & 1);
        if (tid < 16) s_modes[tid] = 0;
        xa_wave_sync();
        if (mineEligible && myRank < maxCand) s_modes[myRank] = (uint8_t)tid;
        xa_wave_sync();
        if (numEligible <= maxCand)
        {
            if (tid == 0) { s_num = numEligible; po->num_cand = (uint32_t)numEligible; }
        }
        else
        {
            /* updateCandList for the rest, the list across lanes 0..15 (one place per lane; places beyond maxCand hold 0 and are never the largest): per
             * eligible mode one 16-lane maximum with row shifts (DPP: no LDS traffic), the first lane holding it is the place to replace.  (A single lane
             * doing the same compares one after the other costs five microseconds.) */
            uint32_t myMode = tid < 16 ? s_modes[tid] : 0;
            const uint32_t got = (uint32_t)__shfl((int)myCost, (int)myMode, 64);         /* the cost of the mode in this lane's place */
            uint32_t mine = tid < maxCand ? got : 0;
            /* drop the modes already placed */
            for (int k = 0; k < maxCand; k++) todo &= todo - 1;
#define XA_ROW_SHR(v, n) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), 0x110 + (n), 0xF, 0xF, true))
            while (todo)
            {
                const int m = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)myCost, m);
                uint32_t cur = mine, o;
                o = XA_ROW_SHR(cur, 1); cur = o > cur ? o : cur;
                o = XA_ROW_SHR(cur, 2); cur = o > cur ? o : cur;
                o = XA_ROW_SHR(cur, 4); cur = o > cur ? o : cur;
                o = XA_ROW_SHR(cur, 8); cur = o > cur ? o : cur;
                const uint32_t maxValue = (uint32_t)__builtin_amdgcn_readlane((int)cur, 15);
                const unsigned long long holders = __ballot(tid < 16 && mine == maxValue);
                const int maxIndex = holders ? __ffsll((long long)holders) - 1 : 0;
                if (c < maxValue && tid == maxIndex) { mine = c; myMode = (uint32_t)m; }
            }
#undef XA_ROW_SHR
            if (tid < 16) s_modes[tid] = (uint8_t)myMode;
            if (tid == 0) { s_num = maxCand; po->num_cand = (uint32_t)maxCand; }
        }
        xa_wave_sync();
        if (tid < 16) po->modes[tid] = tid < s_num ? s_modes[tid] : 0;
    }
    __syncthreads();
    XA_STAGE(13);
    TuLds& s = reinterpret_cast<TuLds*>(smem)[wv];
    IntraTuLds& ip = reinterpret_cast<IntraTuLds*>(smem + nwv * sizeof(TuLds))[wv];
    const int n = s_num;
    for (int i = wv; i < n; i += nwv)
    {
        x265amd_intra_tu_job J = P.tmpl;
        J.tu.dir_mode = s_modes[i];
        J.tu.pred += (uint64_t)i * P.slot_pixels * sizeof(pixel); J.tu.recon += (uint64_t)i * P.slot_pixels * sizeof(pixel);
        J.tu.coeff += (uint64_t)i * P.slot_coeffs * sizeof(int16_t); J.tu.resi += (uint64_t)i * P.slot_coeffs * sizeof(int16_t);
        wave_intra_tu_chain_body<false>(J, nullptr, res + i, s, ip, nullptr, lane);
    }
}

#endif
