/* The fused prediction-unit step of the intra RD (include/x265amd.h: x265amd_intra_pu): shared by its kernel (tu_kernels.hip) and the job server
 * (device_queue.hip). */
#ifndef X265AMD_INTRA_PU_DEV_H
#define X265AMD_INTRA_PU_DEV_H
#include "tu_dev.h"
#include "intra_dev.h"
#include "entropy_dev.h"

/* nb*: the scan's neighbour arrays, kept for the chains; fenc: its copy of the source block (row length N) -- the candidates' chains read the source five times
 * each (residual, two distortions, two psy energies), from here instead of from global memory (a microsecond per dependent read) */
#ifndef XA_CHROMA_AHEAD_OFF
#define XA_CHROMA_AHEAD_OFF 0           /* 1: an 8x8 2Nx2N CU's chroma decision behind its luma decision, as before round 4's last hours */
#endif
struct IntraPuShared { int32_t sa8d[35]; uint8_t modes[16]; int num; pixel nbRef[136], nbFlt[136]; pixel fenc[32 * 32]; };

/* the candidate list of a prediction unit from its 35 SA8D costs (S.sa8d): called by the first wavefront, all 64 lanes; leaves S.modes / S.num */
XA_DEV void wave0_candidate_list(IntraPuShared& S, uint32_t preds0, uint32_t preds1, uint32_t preds2, uint32_t rbits, uint32_t mpmBase, unsigned long long lambda, int maxCandIn, int tid)
{
    /* estIntraPredQT (search.cpp:1615-1650): costs (a lane per mode), the padded best, then updateCandList over the eligible modes in mode order.  Costs fit
     * 32 bits with room to spare (sa8d of a 32x32 block of 12-bit samples < 2^28, bits x lambda >> 8 < 2^14), so the comparisons are the reference's. */
    if (tid < 64)
    {
        const uint32_t kMax = 0xFFFFFFFFu;
        uint32_t myCost = kMax;
        if (tid < 35)
        {
            uint32_t b = rbits;
            if (tid == preds0) b = mpmBase + 1u;
            else if (tid == preds1 || tid == preds2) b = mpmBase + 2u;
            myCost = (uint32_t)S.sa8d[tid] + (uint32_t)(((unsigned long long)b * lambda + 128) >> 8);
        }
        uint32_t bcost = myCost;
        for (int off = 32; off; off >>= 1) { const uint32_t o = __shfl_xor(bcost, off, 64); bcost = o < bcost ? o : bcost; }
        const uint32_t padded = bcost + (bcost >> 2);
        unsigned long long todo = __ballot(tid < 35 && (myCost < padded || tid == preds0));
        const int maxCand = maxCandIn > 16 ? 16 : maxCandIn;
        const int numEligible = __popcll(todo);
        /* the first maxCand eligible modes take the places in order (each replaces the first empty place) */
        const int myRank = __popcll(todo & ((1ull << tid) - 1));
        const bool mineEligible = tid < 35 && ((todo >> tid) & 1);
        if (tid < 16) S.modes[tid] = 0;
        xa_wave_sync();
        if (mineEligible && myRank < maxCand) S.modes[myRank] = (uint8_t)tid;
        xa_wave_sync();
        if (numEligible <= maxCand)
        {
            if (tid == 0) { S.num = numEligible; }
        }
        else
        {
            /* updateCandList for the rest, the list across lanes 0..15 (one place per lane; places beyond maxCand hold 0 and are never the largest): per
             * eligible mode one 16-lane maximum with row shifts (DPP: no LDS traffic), the first lane holding it is the place to replace.  (A single lane
             * doing the same compares one after the other costs five microseconds.) */
            uint32_t myMode = tid < 16 ? S.modes[tid] : 0;
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
            if (tid < 16) S.modes[tid] = (uint8_t)myMode;
            if (tid == 0) { S.num = maxCand; }
        }
        xa_wave_sync();
    }
}

/* x265amd_intra_pu: scan, candidate list, candidate chains by one workgroup (see include/x265amd.h).  smem: the larger of IntraScanLds and one
 * TuLds + IntraTuLds per wavefront. */
XA_DEV void block_intra_pu(const x265amd_intra_pu_job* pj, x265amd_intra_pu_out* po, x265amd_tu_result* res, char* smem, int tid, int nthr)
{
    __shared__ IntraPuShared S;
    XA_STAGE(15);
    const x265amd_intra_pu_job P = xa_ld_record(pj);
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    XA_STAGE(13);
    {
        x265amd_intra_job sj;
        sj.recon = P.tmpl.nb; sj.fenc = P.tmpl.tu.fenc; sj.avail = P.tmpl.avail; sj.recon_stride = P.tmpl.nb_stride; sj.fenc_stride = P.tmpl.tu.fenc_stride;
        sj.log2_tr_size = P.tmpl.tu.log2_tr_size; sj.strong_smoothing = P.tmpl.strong_smoothing;
        block_intra_scan_job(sj, S.sa8d, nullptr, *reinterpret_cast<IntraScanLds*>(smem), tid, nthr);
    }
    __syncthreads();
    XA_STAGE(14);
    {
        const IntraScanLds& sc = *reinterpret_cast<const IntraScanLds*>(smem);
        const int n4 = 4 << P.tmpl.tu.log2_tr_size;
        for (int i = tid; i <= n4; i += nthr) { S.nbRef[i] = sc.ref[i]; S.nbFlt[i] = sc.flt[i]; }
        for (int i = tid; i < (1 << (2 * P.tmpl.tu.log2_tr_size)); i += nthr) S.fenc[i] = sc.fenc[i];
    }
    if (tid < 35) po->sa8d[tid] = S.sa8d[tid];
    if (tid < 64)
    {
        wave0_candidate_list(S, P.preds[0], P.preds[1], P.preds[2], P.rbits, P.mpm_base, P.lambda, P.max_cand, tid);
        if (tid == 0) po->num_cand = (uint32_t)S.num;
        if (tid < 16) po->modes[tid] = tid < S.num ? S.modes[tid] : 0;
    }
    __syncthreads();
    XA_STAGE(13);
    TuLds& s = reinterpret_cast<TuLds*>(smem)[wv];
    IntraTuLds& ip = reinterpret_cast<IntraTuLds*>(smem + nwv * sizeof(TuLds))[wv];
    const int n = S.num;
    for (int i = wv; i < n; i += nwv)
    {
        x265amd_intra_tu_job J = P.tmpl;
        J.tu.dir_mode = S.modes[i];
        J.tu.fenc = (uint64_t)(uintptr_t)(const void*)S.fenc; J.tu.fenc_stride = 1 << P.tmpl.tu.log2_tr_size;
        J.tu.pred += (uint64_t)i * P.slot_pixels * sizeof(pixel); J.tu.recon += (uint64_t)i * P.slot_pixels * sizeof(pixel);
        J.tu.coeff += (uint64_t)i * P.slot_coeffs * sizeof(int16_t); J.tu.resi += (uint64_t)i * P.slot_coeffs * sizeof(int16_t);
        wave_intra_tu_chain_body<false>(J, nullptr, res + i, s, ip, nullptr, lane, S.nbRef, S.nbFlt);
    }
}

/* ---- a CU as a link of a device-run chain (include/x265amd.h: x265amd_intra_nxn_job.chain) ---- */
#define XA_CHAIN_TIMEOUT_TICKS 200000000ll         /* two seconds of the 100 MHz clock: a chain that stands this long is broken, and the host is told rather than left waiting */

/* waits (one lane) until *word >= want; false when it gives up */
XA_DEV bool xa_chain_wait(const uint64_t* word, uint64_t want)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want)
    {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > XA_CHAIN_TIMEOUT_TICKS) return false;
    }
    /* SYSTEM scope, on both sides: with agent-scope fences here the two workgroups of a chained CU (usually on different XCDs) lost each other's writes whenever another
     * kernel ran atomics on the device at the same time -- eleven runs out of twelve under a flood of them, one in five beside the first form of the weight-cost kernel
     * (X265AMD_WP_FLOOD=1,11040 with dbg/wp_md5.py reproduces it when this says "agent").  Measured cost: none outside the noise */
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __builtin_amdgcn_s_dcache_inv();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return true;
}
/* everything this workgroup has written is visible to the other before `word` changes (called by one lane after a barrier that follows the writes) */
XA_DEV void xa_chain_publish(uint64_t* word, uint64_t value)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          /* system scope: see xa_chain_wait */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* the command's start as a link: wait for the CU before it, then take contexts, fraction and neighbour modes from the chain (sP: the record's copy in LDS).
 * Returns false when the wait was given up. */
XA_DEV bool nxn_chain_begin(x265amd_intra_nxn_job& sP, int tid, int nthr)
{
    __shared__ int s_chainOk;
    x265amd_intra_chain* ch = reinterpret_cast<x265amd_intra_chain*>(sP.chain);
    XA_LINK_START();
    if (sP.chain_first) return true;
    XA_CHAIN_START();
    if (tid == 0) s_chainOk = xa_chain_wait(&ch->seq, sP.chain_token) ? 1 : 0;
    if (sP.chain_role != 1) XA_LINK_T(7);
    __syncthreads();
    XA_CHAIN(sP.chain_role == 1 ? 0 : 1);
    if (!s_chainOk) return false;
    for (int i = tid; i < X265AMD_CTX_STRIDE; i += nthr) sP.ctx[i] = ch->ctx[i];
    if (tid == 0)
    {
        /* (the record's mode fields are read and written by this lane alone) */
        const uint64_t frac = ch->frac & 32767;
        uint8_t m[4];
        for (int i = 0; i < 4; i++)
        {
            const uint8_t src = sP.mode_src[i];
            m[i] = src == 0xFF ? (i < 2 ? sP.left_mode[i] : sP.above_mode[i - 2]) : ch->mode[src >> 2][src & 3];
        }
        sP.left_mode[0] = m[0]; sP.left_mode[1] = m[1]; sP.above_mode[0] = m[2]; sP.above_mode[1] = m[3];
        sP.scan_frac = (uint32_t)frac;
        /* what codeIntraLumaQT finds in front of the first unit's direction in an I slice: the partition size bin (2Nx2N: 1, NxN: 0) */
        const uint32_t bin = (sP.num_units == 0 || sP.num_units == 4) ? 0u : 1u;
        sP.frac_start[0] = frac + en_bits[ch->ctx[8] ^ bin];
        sP.frac_start[1] = sP.frac_start[2] = sP.frac_start[3] = frac;
    }
    __syncthreads();
    return true;
}

#include "intra_cu_dev.h"
#include "intra_nxn4_dev.h"

/* x265amd_intra_nxn (include/x265amd.h): the four 4x4 prediction units of an 8x8 NxN CU, decisions included, by one workgroup. */
/* smemBytes: the dynamic LDS behind smem (with RDOQ the wavefronts that run chains side by side are as many as fit with their RDOQ areas) */
XA_DEV void block_intra_nxn(const x265amd_intra_nxn_job* pj, x265amd_intra_nxn_out* po, char* smem, int tid, int nthr, int smemBytes)
{
    __shared__ IntraPuShared S;
    __shared__ x265amd_tu_result s_res[16];
    __shared__ unsigned long long s_cost[16];
    __shared__ uint8_t s_ctxw[16][X265AMD_CTX_STRIDE];
    __shared__ int s_win;
    __shared__ uint8_t s_winMode[4];
    __shared__ uint32_t s_pickSa8d;
    __shared__ uint8_t s_preds0[4];                 /* a chained CU (role 2): the first unit's predictors and the chroma modes' fractions, for the CU's own bit count */
    __shared__ unsigned long long s_cfrac[5];
    __shared__ uint32_t s_numSig0;
    /* a chained CU decided in this general form (role 1 with RDOQ: the sixteen-lane form of intra_nxn4_dev.h has no rdoQuant): what the decision needs of the four units */
    __shared__ uint8_t s_predsAll[4][3];
    __shared__ x265amd_tu_result s_ures[4];
    __shared__ int16_t s_lev[64];
    __shared__ uint32_t s_psyNxn, s_resNxn;
    XA_STAGE(15);
    XA_NXN_START(0);
    __shared__ x265amd_intra_nxn_job sP;           /* the job record: about a kilobyte, indexed by the unit -- in LDS, not in registers */
    static_assert(sizeof(x265amd_intra_nxn_job) % 8 == 0, "job records are sequences of 64-bit words");
    __syncthreads();
    for (int i = tid; i < (int)(sizeof(x265amd_intra_nxn_job) / 8); i += nthr)
        reinterpret_cast<uint64_t*>(&sP)[i] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(pj) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    /* the estimator's tables beside it: the lanes that count bits look them up bin after bin.  (Filled BEFORE a chained CU waits for the CU in front of it: the other
     * evaluation of a chained CU has nothing else to do then, and what it does after the wait is the chain's critical path.) */
    __shared__ uint32_t s_enBits[128];
    __shared__ uint8_t s_enLps[64];
    __shared__ uint32_t s_step[256];               /* bits and next state per (state, bin): the one look-up of the per-context walks (wave_coeff_bits_4x4) */
    const bool nxn4Form = (sP.num_units == 0 || sP.num_units == 4) && (sP.unit_log2 == 0 || sP.unit_log2 == 2) && !sP.pick_sa8d && !sP.no_picture && (!sP.rdoq_level || !sP.rdoq_general);
    if (!nxn4Form)
    {
        if (tid < 128) s_enBits[tid] = en_bits[tid];
        if (tid < 64) s_enLps[tid] = en_lpsNext[tid];
        for (int i = tid; i < 256; i += nthr) s_step[i] = en_step.v[i];
    }
    if (sP.chain)
    {
        if (!nxn_chain_begin(sP, tid, nthr))
        {
            /* the chain stands: say so where the host looks, and let whoever waits for this command see it */
            if (tid == 0)
            {
                if (sP.chain_role == 1) reinterpret_cast<x265amd_intra_cu8_result*>(sP.cu_out)->status = 2;
                else xa_chain_publish(&reinterpret_cast<x265amd_intra_peer*>(sP.peer)->ready, ~0ull);
            }
            return;
        }
        if (sP.chain_role == 2) po = &reinterpret_cast<x265amd_intra_peer*>(sP.peer)->out;
    }
    if (nxn4Form)
    {
        /* the NxN CU proper: its own form, nothing but LDS and registers between the first and the last instruction (intra_nxn4_dev.h) */
        block_intra_nxn4(sP, po, *reinterpret_cast<Nxn4Lds*>(smem), S, tid, nthr, smem + ((sizeof(Nxn4Lds) + 15) & ~(size_t)15));
        return;
    }
    const EnTabs tabs{ s_enBits, s_enLps };
    const x265amd_intra_nxn_job& P = sP;
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
    const uint32_t adi = P.ctx[13];                                                         /* C_ADI: prev_intra_luma_pred_flag */
    const uint32_t rbits = (uint32_t)(((unsigned long long)P.scan_frac + en_bits[adi ^ 0]) >> 15) + 5;
    const uint32_t mpmBase = (uint32_t)(((unsigned long long)P.scan_frac + en_bits[adi ^ 1]) >> 15);
    const int maxCand = P.max_cand > 16 ? 16 : P.max_cand;
    const int numUnits = P.num_units ? P.num_units : 4, unitLog2 = P.unit_log2 ? P.unit_log2 : 2, N = 1 << unitLog2;
    const int cbfCtx = CTX_QT_CBF + (numUnits == 1 ? 1 : 0);                               /* C_QT_CBF + !tuDepth */
    XA_NXN_START(numUnits == 1 ? unitLog2 - 2 : 0);          /* (the record's load goes to kind 0 / stage 0 .. never mind: it is counted below from here) */
    XA_NXN(0);
    /* An 8x8 CU coded 2Nx2N: its chroma decision (one 4x4 block per plane, five listed modes) needs the luma decision only for the DERIVED mode, and four of the five
     * listed modes are always among planar, vertical, horizontal, DC and 34.  Those five are evaluated by the two wavefronts the luma candidates leave idle (at most six
     * candidates), BESIDE the luma candidates' chains; behind the luma decision what is left is the signalling's share of the bits (which mode is the derived one) and,
     * when the luma mode is none of the five, one more evaluation.  The chroma working set lies behind the luma chains' LDS instead of on top of it. */
    static_assert(XA_WAVE == 64, "");
    /* RDOQ: the two bit-estimate tables of the command (luma units, chroma blocks) from its contexts, and how many wavefronts fit with an RDOQ area each */
    const bool rdoq = P.rdoq_level != 0;
    __shared__ int32_t s_est[2][184];
    const int cLog2r = numUnits == 1 ? unitLog2 - 1 : 2;
    const int perWave = (int)(sizeof(TuLds) + sizeof(IntraTuLds));
    const int rqBytesL = rdoq_ref_bytes(unitLog2), rqBytesC = rdoq_ref_bytes(cLog2r);
    const int nwvL = rdoq ? min(nwv, smemBytes / (perWave + rqBytesL)) : nwv, nwvC = rdoq ? min(nwv, smemBytes / (perWave + rqBytesC)) : nwv;
    if (rdoq)
    {
        for (int i = tid; i < 2 * 184; i += nthr) (&s_est[0][0])[i] = 0;
        __syncthreads();
        if (wv == 0) wave_est_bit(P.ctx, s_est[0], unitLog2, 1, lane);
        else if (wv == 1) wave_est_bit(P.ctx, s_est[1], cLog2r, 0, lane);
        __syncthreads();
    }
    const RdoqParams rpL = { s_est[0], P.rdoq_lambda2[0], P.rdoq_lambda[0], P.psy_rdoq_scale, P.rdoq_level, P.rdoq_tu_depth };
    const bool chromaAhead = P.do_chroma && numUnits == 1 && unitLog2 == 3 && !P.pick_sa8d && nwv >= 8 && maxCand <= 6 && !XA_CHROMA_AHEAD_OFF && !rdoq;
    Nxn4Lds& S4c = *reinterpret_cast<Nxn4Lds*>(smem + (chromaAhead ? (size_t)nwv * (sizeof(TuLds) + sizeof(IntraTuLds)) : 0));
    __shared__ uint8_t s_specModes[8];
    if (tid < 8) { const uint8_t fixed5[8] = { 0, 26, 10, 1, 34, 0, 0, 0 }; s_specModes[tid] = fixed5[tid]; }
    for (int k = 0; k < numUnits; k++)
    {
        const x265amd_intra_tu_job& T = P.tmpl[k];
        /* getIntraDirLumaPredictor: the left / above units' modes */
        __syncthreads();
        const uint32_t left = (k & 1) ? s_winMode[k - 1] : P.left_mode[k >> 1], above = (k & 2) ? s_winMode[k - 2] : P.above_mode[k & 1];
        uint32_t p0, p1, p2;
        if (left == above)
        {
            if (left >= 2) { p0 = left; p1 = ((left - 2 + 31) & 31) + 2; p2 = ((left - 2 + 1) & 31) + 2; }
            else { p0 = 0; p1 = 1; p2 = 26; }
        }
        else { p0 = left; p1 = above; p2 = (left && above) ? 0 : ((left + above) < 2 ? 26 : 1); }
        if (k == 0 && tid == 0) { s_preds0[0] = (uint8_t)p0; s_preds0[1] = (uint8_t)p1; s_preds0[2] = (uint8_t)p2; }
        if (tid == 0) { s_predsAll[k & 3][0] = (uint8_t)p0; s_predsAll[k & 3][1] = (uint8_t)p1; s_predsAll[k & 3][2] = (uint8_t)p2; }
        XA_STAGE(16);
        XA_NXN(1);
        {
            x265amd_intra_job sj;
            sj.recon = T.nb; sj.fenc = T.tu.fenc; sj.avail = T.avail; sj.recon_stride = T.nb_stride; sj.fenc_stride = T.tu.fenc_stride;
            sj.log2_tr_size = T.tu.log2_tr_size; sj.strong_smoothing = T.strong_smoothing;
            block_intra_scan_job(sj, S.sa8d, nullptr, *reinterpret_cast<IntraScanLds*>(smem), tid, nthr);
        }
        __syncthreads();
        XA_STAGE(17);
        XA_NXN(2);
        {
            const IntraScanLds& sc = *reinterpret_cast<const IntraScanLds*>(smem);
            for (int i = tid; i <= 4 * N; i += nthr) { S.nbRef[i] = sc.ref[i]; S.nbFlt[i] = sc.flt[i]; }
            for (int i = tid; i < N * N; i += nthr) S.fenc[i] = sc.fenc[i];
        }
        if (chromaAhead && wv >= 6)
        {
            /* the chroma working set, by the two wavefronts that have nothing to do while the first one makes the candidate list */
            const int t = tid - 6 * XA_WAVE;
            nxn4_fill_tabs(S4c.tb, t);
            S4c.enBits[t] = s_enBits[t];
            if (t < 64) S4c.enLps[t] = s_enLps[t];
            S4c.step[t] = s_step[t]; S4c.step[t + 128] = s_step[t + 128];
            if (wv == 7)
                for (int pl = 0; pl < 2; pl++)
                    (void)nxn4_neighbours(reinterpret_cast<const pixel*>(P.ctmpl[pl].nb), (int)P.ctmpl[pl].nb_stride, (uint32_t)P.ctmpl[pl].avail, S4c.cref[pl], S4c.csw[pl], lane);
            else if (lane < 32)
            {
                const int pl = lane >> 4, i = lane & 15;
                S4c.cfenc[pl][i] = reinterpret_cast<const pixel*>(P.ctmpl[pl].tu.fenc)[(i >> 2) * P.ctmpl[pl].tu.fenc_stride + (i & 3)];
            }
        }
        if (tid < 64)
        {
            if (P.pick_sa8d)
            {
                /* checkIntraInInter: cost = sa8d + mode bits as in the list; the first minimum in the order DC, planar, 2..34 */
                unsigned long long key = ~0ull;
                if (tid < 35)
                {
                    uint32_t b = rbits;
                    if ((uint32_t)tid == p0) b = mpmBase + 1u;
                    else if ((uint32_t)tid == p1 || (uint32_t)tid == p2) b = mpmBase + 2u;
                    const uint32_t c = (uint32_t)S.sa8d[tid] + (uint32_t)(((unsigned long long)b * P.lambda + 128) >> 8);
                    const uint32_t rank = tid == 1 ? 0u : (tid == 0 ? 1u : (uint32_t)tid);
                    key = ((unsigned long long)c << 8) | rank;
                }
                for (int off = 32; off; off >>= 1) { const unsigned long long o = __shfl_xor(key, off, 64); key = o < key ? o : key; }
                const uint32_t rank = (uint32_t)(key & 255u), mode = rank == 0 ? 1u : (rank == 1 ? 0u : rank);
                if (tid == 0) { S.modes[0] = (uint8_t)mode; S.num = 1; s_pickSa8d = (uint32_t)S.sa8d[mode]; }
                xa_wave_sync();
            }
            else wave0_candidate_list(S, p0, p1, p2, rbits, mpmBase, P.lambda, maxCand, tid);
        }
        __syncthreads();
        XA_STAGE(18);
        XA_NXN(3);
        const int n = S.num;
        TuLds& s = reinterpret_cast<TuLds*>(smem)[wv < nwvL ? wv : 0];
        IntraTuLds& ip = reinterpret_cast<IntraTuLds*>(smem + nwvL * sizeof(TuLds))[wv < nwvL ? wv : 0];
        RdoqRef rrL = rdoq_ref_at(smem + (size_t)nwvL * perWave + (size_t)(wv < nwvL ? wv : 0) * rqBytesL, unitLog2);
        rrL.est = s_est[0];             /* (the command's table is in LDS already: read where it lies) */
        if (chromaAhead && wv >= 6)
        {
            /* U, then V on the contexts U has moved: modes 0 .. 3 of the five by the groups of wavefront 6, the fifth by the first group of wavefront 7 */
            const EnTabs tabsC{ S4c.enBits, S4c.enLps };
            nxn4_chroma_spec(P, S4c, s_specModes, (wv - 6) * 4, 5, 0, tabsC, lane, lane >> 4, lane & 15);
            xa_wave_sync();
            nxn4_chroma_spec(P, S4c, s_specModes, (wv - 6) * 4, 5, 1, tabsC, lane, lane >> 4, lane & 15);
        }
        for (int i = wv; i < n && wv < nwvL && !(chromaAhead && wv >= 6); i += nwvL)
        {
            x265amd_intra_tu_job J = T;
            const uint32_t mode = S.modes[i];
            J.tu.dir_mode = (uint8_t)mode;
            J.tu.fenc = (uint64_t)(uintptr_t)(const void*)S.fenc; J.tu.fenc_stride = N;
            J.tu.pred += (uint64_t)i * P.slot_pixels * sizeof(pixel); J.tu.recon += (uint64_t)i * P.slot_pixels * sizeof(pixel);
            J.tu.coeff += (uint64_t)i * P.slot_coeffs * sizeof(int16_t); J.tu.resi += (uint64_t)i * P.slot_coeffs * sizeof(int16_t);
            if (rdoq) wave_intra_tu_chain_body<true>(J, nullptr, &s_res[i], s, ip, nullptr, lane, S.nbRef, S.nbFlt, &rrL, &rpL);
            else wave_intra_tu_chain_body<false>(J, nullptr, &s_res[i], s, ip, nullptr, lane, S.nbRef, S.nbFlt);
            XA_STAGE(19);
            /* the candidate's bits and cost (codeIntraLumaQT, search.cpp:357-400).  The levels are still in this wavefront's LDS.  A 4x4 unit: its contexts as
             * independent state machines across the lanes, straight from the job's context set (a candidate is only priced, nothing is written back) */
            xa_wave_sync();
            const uint32_t numSig = s_res[i].num_sig;
            unsigned long long coeffFrac = 0;
            if (numSig) coeffFrac = unitLog2 == 2 ? wave_coeff_bits_4x4(P.ctx, nullptr, s.q, 0, 1, (int)mode, T.tu.sign_hide, s_step, lane)
                                                  : wave_coeff_bits(P.ctx, nullptr, s.q, unitLog2, 0, 1, (int)mode, T.tu.sign_hide, s_step, lane);
            if (lane == 0)
            {
                const uint8_t* cw = P.ctx;
                const int pidx = mode == p0 ? 0 : (mode == p1 ? 1 : (mode == p2 ? 2 : -1));
                unsigned long long frac = P.frac_start[k];
                frac += s_enBits[cw[13] ^ (pidx != -1 ? 1u : 0u)];
                frac += (unsigned long long)(pidx != -1 ? 1 + (pidx != 0) : 5) << 15;
                const x265amd_tu_result r = s_res[i];
                frac += s_enBits[cw[cbfCtx] ^ (r.num_sig != 0 ? 1u : 0u)];
                frac += coeffFrac;
                const unsigned long long bits = (uint32_t)(frac >> 15);
                const unsigned long long dist = r.nz_dist;
                s_cost[i] = P.psy_scale ? dist + ((P.psy_scale * (unsigned long long)r.nz_energy) >> 24) + ((bits * P.lambda2) >> 8) : dist + ((bits * P.lambda2 + 128) >> 8);
            }
        }
        XA_STAGE(20);
        XA_NXN(4);
        __syncthreads();
        XA_NXN(5);
        if (tid == 0)
        {
            int w = 0;
            unsigned long long best = ~0ull;
            for (int i = 0; i < n; i++) if (s_cost[i] < best) { best = s_cost[i]; w = i; }
            s_win = w; s_winMode[k] = S.modes[w];
            po->mode[k] = S.modes[w]; po->num_cand[k] = (uint8_t)n; po->res[k] = s_res[w];
            if (k == 0) s_numSig0 = s_res[w].num_sig;
            s_ures[k & 3] = s_res[w];
        }
        __syncthreads();
        XA_NXN(6);
        {
            /* the winner's blocks: reconstruction into the picture and the layer tile, prediction into the prediction tile; its levels to the host */
            const int w = s_win;
            const pixel* rec = reinterpret_cast<const pixel*>(T.tu.recon) + (size_t)w * P.slot_pixels;
            const pixel* prd = reinterpret_cast<const pixel*>(T.tu.pred) + (size_t)w * P.slot_pixels;
            const int16_t* lv = reinterpret_cast<const int16_t*>(T.tu.coeff) + (size_t)w * P.slot_coeffs;
            int16_t* lvOut = P.levels_dst ? reinterpret_cast<int16_t*>(P.levels_dst) : &po->levels[0][0];
            for (int t = tid; t < N * N; t += nthr)
            {
                const int y = t >> unitLog2, x = t & (N - 1);
                const pixel v = rec[y * T.tu.recon_stride + x];
                if (!P.no_picture) reinterpret_cast<pixel*>(T.nb)[(long)y * T.nb_stride + x] = v;
                reinterpret_cast<pixel*>(P.layer_dst[k])[y * 64 + x] = v;
                if (P.recon_dst[k]) reinterpret_cast<pixel*>(P.recon_dst[k])[y * 64 + x] = v;
                reinterpret_cast<pixel*>(P.pred_dst[k])[y * 64 + x] = prd[y * T.tu.pred_stride + x];
                lvOut[k * 16 + t] = lv[t];
                if (numUnits == 4) s_lev[(k * 16 + t) & 63] = lv[t];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            /* the next unit's neighbours */
        }
        XA_STAGE(21);
        XA_NXN(7);
    }
    /* the CU's luma measurements on the finished 8x8 block: psy energy of the reconstruction, residual energy of the prediction */
    __syncthreads();
    if (wv == 0 && numUnits == 4)
    {
        const x265amd_intra_tu_job& T0 = P.tmpl[0];
        const pixel* f = reinterpret_cast<const pixel*>(T0.tu.fenc);
        const int psy = wave_psy_cost(f, T0.tu.fenc_stride, reinterpret_cast<const pixel*>(T0.nb), T0.nb_stride, 1, lane);
        const uint64_t sse = wave_sse_pp(f, T0.tu.fenc_stride, reinterpret_cast<const pixel*>(P.pred_dst[0]), 64, 8, lane);
        if (lane == 0) { po->psy_energy = (uint32_t)psy; po->res_energy = (uint32_t)sse; s_psyNxn = (uint32_t)psy; s_resNxn = (uint32_t)sse; }
    }
    XA_NXN(8);
    if (!P.do_chroma) return;
    /* ---- estIntraPredChromaQT for the one block per plane (4x4 for an 8x8 CU, N/2 for a larger single unit): a wavefront per mode ---- */
    __shared__ x265amd_tu_result s_cres[5][2];
    __shared__ uint8_t s_cmode[5];
    __shared__ pixel s_cfenc[2][16];                /* the two source blocks of the 4x4 case, read by the five modes' chains */
    const int cLog2 = numUnits == 1 ? unitLog2 - 1 : 2, CN = 1 << cLog2;
    const uint32_t lumaDir = s_winMode[0];
    if (chromaAhead && P.chain && P.chain_role == 2)
    {
        /* The other evaluation of a chained CU is what the chain waits for (the deciding command stands five microseconds per CU: XA_LINK), so its tail is ONE barrier
         * interval instead of six with the CU's bit count behind them: the last wavefront counts the luma share of the CU's bits (Search::checkIntra's count at its end,
         * search.cpp:1254-1275: partition size, prediction info, the coded block flag and the coefficients on a copy of the start contexts) while the first one makes
         * the chroma decision of the branch below with wavefront-level fences; then the sums, the merged contexts, and the word the deciding command waits for. */
        __shared__ uint8_t s_ftSrc[5], s_ftC14[5];
        __shared__ unsigned long long s_ftFrac, s_ftMv;
        Nxn4Lds& S4 = S4c;
        x265amd_intra_peer* peer = reinterpret_cast<x265amd_intra_peer*>(P.peer);
        uint8_t* run = s_ctxw[5];
        if (wv == nwv - 1)
        {
            for (int b = lane; b < X265AMD_CTX_STRIDE; b += XA_WAVE) run[b] = P.ctx[b];
            xa_wave_sync();
            const uint32_t mode = s_winMode[0];
            const uint32_t numSigY = s_numSig0;
            unsigned long long frac = P.scan_frac, mvf = 0;
            if (lane == 0)
            {
                frac += cb_bin_t(tabs, run + 8, 1u);                                                   /* C_PART_SIZE: 2Nx2N */
                const int pidx = mode == s_preds0[0] ? 0 : (mode == s_preds0[1] ? 1 : (mode == s_preds0[2] ? 2 : -1));
                frac += cb_bin_t(tabs, run + 13, pidx != -1 ? 1u : 0u);
                frac += (unsigned long long)(pidx != -1 ? 1 + (pidx != 0) : 5) << 15;
                mvf = frac;                                                                            /* (the chroma mode's share joins below) */
                frac += cb_bin_t(tabs, run + CTX_QT_CBF + 1, numSigY != 0 ? 1u : 0u);
            }
            xa_wave_sync();
            frac = __shfl(frac, 0, 64); mvf = __shfl(mvf, 0, 64);
            if (numSigY) frac += wave_coeff_bits(run, run, P.levels_dst ? reinterpret_cast<const int16_t*>(P.levels_dst) : &po->levels[0][0], unitLog2, 0, 1, (int)mode, P.tmpl[0].tu.sign_hide, s_step, lane);
            xa_wave_sync();
            if (lane == 0) { s_ftFrac = frac; s_ftMv = mvf; }
        }
        else if (wv == 0)
        {
            int src = 0;
            uint32_t listed = 0;
            if (lane < 5)
            {
                uint32_t list[5] = { 0, 26, 10, 1, 36 };                /* CUData::getAllowedChromaDir (cudata.cpp:889-907) */
                for (int i = 0; i < 4; i++) if (lumaDir == list[i]) { list[i] = 34; break; }
                listed = list[lane];
                s_cmode[lane] = (uint8_t)listed;
                const uint32_t mode = listed == 36 ? lumaDir : listed;
                src = mode == 0 ? 0 : (mode == 26 ? 1 : (mode == 10 ? 2 : (mode == 1 ? 3 : (mode == 34 ? 4 : 5))));
                s_ftSrc[lane] = (uint8_t)src;
            }
            if (lane == 0) s_specModes[5] = (uint8_t)lumaDir;
            const bool need = __ballot(lane < 5 && src == 5) != 0;
            xa_wave_sync();
            const EnTabs tabsC{ S4.enBits, S4.enLps };
            if (need)
            {
                nxn4_chroma_spec(P, S4, s_specModes, 5, 6, 0, tabsC, lane, lane >> 4, lane & 15);
                xa_wave_sync();
                nxn4_chroma_spec(P, S4, s_specModes, 5, 6, 1, tabsC, lane, lane >> 4, lane & 15);
                xa_wave_sync();
            }
            if (lane < 5)
            {
                uint8_t c14 = P.ctx[14];
                unsigned long long frac = P.scan_frac;
                frac += cb_bin_t(tabs, &c14, listed == 36 ? 0u : 1u);                                   /* C_CHROMA_PRED (codeIntraDirChroma, entropy.cpp:1644-1664) */
                if (listed != 36) frac += 2ull << 15;
                frac += S4.cfrac[src];                                                                  /* the coded block flags and the coefficients: the slot's share */
                const x265amd_tu_result rU = S4.cres[src][0], rV = S4.cres[src][1];
                const unsigned long long dist = rU.nz_dist + rV.nz_dist, energy = (unsigned long long)rU.nz_energy + rV.nz_energy;
                const unsigned long long bits = (uint32_t)(frac >> 15);
                s_cfrac[lane] = frac; s_ftC14[lane] = c14;
                S4.ccost[lane] = P.psy_scale ? dist + ((P.psy_scale * energy) >> 24) + ((bits * P.lambda2) >> 8) : dist + ((bits * P.lambda2 + 128) >> 8);
            }
            xa_wave_sync();
            int w = 0;
            {
                unsigned long long best = ~0ull;
                for (int i = 0; i < 5; i++) if (S4.ccost[i] < best) { best = S4.ccost[i]; w = i; }
            }
            const int sw = s_ftSrc[w], slast = s_ftSrc[4];
            if (lane == 0)
            {
                s_win = w;
                po->chroma_best = (uint32_t)w; po->chroma_reserved = 0;
                po->cres[0] = S4.cres[sw][0]; po->cres[1] = S4.cres[sw][1];
            }
            for (int b = lane; b < X265AMD_CTX_STRIDE; b += XA_WAVE) s_ctxw[w][b] = b == 14 ? s_ftC14[w] : S4.ctxw[sw][b];
            if (lane < 32)
            {
                const int pl = lane >> 4, i = lane & 15, y = i >> 2, x = i & 3;
                const x265amd_intra_tu_job& C = P.ctmpl[pl];
                int16_t* clOut = P.clevels_dst ? reinterpret_cast<int16_t*>(P.clevels_dst) : &po->clevels[0][0];
                reinterpret_cast<pixel*>(P.crecon_dst[pl])[y * 32 + x] = S4.crec[sw][pl][i];
                if (!P.no_picture) reinterpret_cast<pixel*>(C.nb)[(long)y * C.nb_stride + x] = S4.crec[slast][pl][i];      /* the last tried mode's samples: the derived mode's */
                clOut[pl * 16 + i] = S4.clev[sw][pl][i];
            }
        }
        __syncthreads();
        if (wv == 0)
        {
            const int cwIdx = s_win;
            const uint32_t listedW = s_cmode[cwIdx];
            /* the chroma decision's contexts over the luma walk's: what it moved is chroma's alone */
            for (int b = lane; b < X265AMD_CTX_STRIDE; b += XA_WAVE)
            {
                const uint8_t c = s_ctxw[cwIdx][b];
                peer->fctx[b] = c != P.ctx[b] ? c : run[b];
            }
            if (lane == 0)
            {
                peer->ffrac = s_ftFrac + (s_cfrac[cwIdx] - P.scan_frac);
                peer->fmv = s_ftMv + s_enBits[P.ctx[14] ^ (listedW == 36 ? 0u : 1u)] + (listedW != 36 ? (2ull << 15) : 0ull);
            }
        }
        __syncthreads();
        if (tid == 0) xa_chain_publish(&peer->ready, P.chain_token + 1);
        XA_LINK_T(3);
        return;
    }
    if (cLog2 == 2 && !chromaAhead && tid >= 64 && tid < 96)
    {
        const int pl = (tid - 64) >> 4, i = tid & 15;
        s_cfenc[pl][i] = reinterpret_cast<const pixel*>(P.ctmpl[pl].tu.fenc)[(i >> 2) * P.ctmpl[pl].tu.fenc_stride + (i & 3)];
    }
    if (tid < 5)
    {
        /* CUData::getAllowedChromaDir (cudata.cpp:889-907) */
        uint32_t list[5] = { 0, 26, 10, 1, 36 };
        for (int i = 0; i < 4; i++) if (lumaDir == list[i]) { list[i] = 34; break; }
        s_cmode[tid] = (uint8_t)list[tid];
    }
    __syncthreads();
    if (cLog2 == 2 && chromaAhead)
    {
        /* the five modes 0 / 26 / 10 / 1 / 34 were evaluated beside the luma candidates (slots 0 .. 4 of S4c): which slot serves which place of the list, the one
         * evaluation that may be missing (a derived mode that is none of the five: slot 5), then per place the signalling's bits on top of the slot's share */
        Nxn4Lds& S4 = S4c;
        __shared__ uint8_t s_csrc[5], s_c14[5];
        __shared__ int s_needExtra;
        if (tid == 0)
        {
            int need = 0;
            for (int j = 0; j < 5; j++)
            {
                const uint32_t listed = s_cmode[j], mode = listed == 36 ? lumaDir : listed;
                const int src = mode == 0 ? 0 : (mode == 26 ? 1 : (mode == 10 ? 2 : (mode == 1 ? 3 : (mode == 34 ? 4 : 5))));
                s_csrc[j] = (uint8_t)src; need |= src == 5;
            }
            s_needExtra = need;
            s_specModes[5] = (uint8_t)lumaDir;
        }
        __syncthreads();
        if (s_needExtra && wv == 0)
        {
            const EnTabs tabsC{ S4.enBits, S4.enLps };
            nxn4_chroma_spec(P, S4, s_specModes, 5, 6, 0, tabsC, lane, lane >> 4, lane & 15);
            xa_wave_sync();
            nxn4_chroma_spec(P, S4, s_specModes, 5, 6, 1, tabsC, lane, lane >> 4, lane & 15);
        }
        __syncthreads();
        if (tid < 5)
        {
            const int src = s_csrc[tid];
            const uint32_t listed = s_cmode[tid];
            uint8_t c14 = P.ctx[14];
            unsigned long long frac = P.scan_frac;
            frac += cb_bin_t(tabs, &c14, listed == 36 ? 0u : 1u);                                   /* C_CHROMA_PRED (codeIntraDirChroma, entropy.cpp:1644-1664) */
            if (listed != 36) frac += 2ull << 15;
            frac += S4.cfrac[src];                                                                  /* the coded block flags and the coefficients: the slot's share */
            const x265amd_tu_result rU = S4.cres[src][0], rV = S4.cres[src][1];
            const unsigned long long dist = rU.nz_dist + rV.nz_dist, energy = (unsigned long long)rU.nz_energy + rV.nz_energy;
            const unsigned long long bits = (uint32_t)(frac >> 15);
            s_cfrac[tid] = frac; s_c14[tid] = c14;
            S4.ccost[tid] = P.psy_scale ? dist + ((P.psy_scale * energy) >> 24) + ((bits * P.lambda2) >> 8) : dist + ((bits * P.lambda2 + 128) >> 8);
        }
        __syncthreads();
        int w = 0;
        {
            unsigned long long best = ~0ull;
            for (int i = 0; i < 5; i++) if (S4.ccost[i] < best) { best = S4.ccost[i]; w = i; }
        }
        const int sw = s_csrc[w], slast = s_csrc[4];
        if (tid == 0)
        {
            s_win = w;
            po->chroma_best = (uint32_t)w; po->chroma_reserved = 0;
            po->cres[0] = S4.cres[sw][0]; po->cres[1] = S4.cres[sw][1];
        }
        for (int b = tid; b < X265AMD_CTX_STRIDE; b += nthr) s_ctxw[w][b] = b == 14 ? s_c14[w] : S4.ctxw[sw][b];
        if (tid >= 64 && tid < 96)
        {
            const int pl = (tid - 64) >> 4, i = tid & 15, y = i >> 2, x = i & 3;
            const x265amd_intra_tu_job& C = P.ctmpl[pl];
            int16_t* clOut = P.clevels_dst ? reinterpret_cast<int16_t*>(P.clevels_dst) : &po->clevels[0][0];
            reinterpret_cast<pixel*>(P.crecon_dst[pl])[y * 32 + x] = S4.crec[sw][pl][i];
            if (!P.no_picture) reinterpret_cast<pixel*>(C.nb)[(long)y * C.nb_stride + x] = S4.crec[slast][pl][i];      /* the last tried mode's samples: the derived mode's */
            clOut[pl * 16 + i] = S4.clev[sw][pl][i];
        }
    }
    else if (cLog2 == 2)
    {
        /* 4x4 chroma blocks (an 8x8 CU): the sixteen-lane chains of intra_nxn4_dev.h, a group per mode -- the dynamic LDS is free, the luma chains are done.  With RDOQ
         * (round 5) the groups quantise through wave_rdo_quant's sixteen-lane form: the chroma table of this command and eight group areas behind the header */
        Nxn4Lds& S4 = *reinterpret_cast<Nxn4Lds*>(smem);
        if (rdoq)
        {
            for (int i = tid; i < 184; i += nthr) S4.est[1][i] = s_est[1][i];
            if (tid == 0) S4.rq = reinterpret_cast<Rq4Area*>(smem + ((sizeof(Nxn4Lds) + 15) & ~(size_t)15));
        }
        nxn4_fill_tabs(S4.tb, tid);
        if (tid < 128) S4.enBits[tid] = s_enBits[tid];
        if (tid < 64) S4.enLps[tid] = s_enLps[tid];
        for (int i = tid; i < 256; i += nthr) S4.step[i] = s_step[i];
        if (wv == nwv - 1)
            for (int pl = 0; pl < 2; pl++)
            {
                const x265amd_intra_tu_job& C = P.ctmpl[pl];
                (void)nxn4_neighbours(reinterpret_cast<const pixel*>(C.nb), (int)C.nb_stride, (uint32_t)C.avail, S4.cref[pl], S4.csw[pl], lane);
                if (lane < 16) S4.cfenc[pl][lane] = s_cfenc[pl][lane];
            }
        __syncthreads();
        uint32_t list[5];
        for (int i = 0; i < 5; i++) list[i] = s_cmode[i];
        nxn4_chroma_modes(P, S4, list, lumaDir, EnTabs{ S4.enBits, S4.enLps }, lane, wv, lane >> 4, lane & 15);
        __syncthreads();
        int w = 0;
        {
            unsigned long long best = ~0ull;
            for (int i = 0; i < 5; i++) if (S4.ccost[i] < best) { best = S4.ccost[i]; w = i; }
        }
        if (tid == 0)
        {
            s_win = w;
            po->chroma_best = (uint32_t)w; po->chroma_reserved = P.pick_sa8d ? s_pickSa8d : 0;
            po->cres[0] = S4.cres[w][0]; po->cres[1] = S4.cres[w][1];
        }
        if (tid < 5) s_cfrac[tid] = S4.cfrac[tid];
        for (int b = tid; b < X265AMD_CTX_STRIDE; b += nthr) s_ctxw[w][b] = S4.ctxw[w][b];
        if (tid >= 64 && tid < 96)
        {
            const int pl = (tid - 64) >> 4, i = tid & 15, y = i >> 2, x = i & 3;
            const x265amd_intra_tu_job& C = P.ctmpl[pl];
            int16_t* clOut = P.clevels_dst ? reinterpret_cast<int16_t*>(P.clevels_dst) : &po->clevels[0][0];
            reinterpret_cast<pixel*>(P.crecon_dst[pl])[y * 32 + x] = S4.crec[w][pl][i];
            if (!P.no_picture) reinterpret_cast<pixel*>(C.nb)[(long)y * C.nb_stride + x] = S4.crec[4][pl][i];
            clOut[pl * 16 + i] = S4.clev[w][pl][i];
        }
    }
    else {
    /* (with RDOQ the five modes need five wavefronts with an RDOQ area each: the largest chroma block is 16x16, of which six fit) */
    if (wv < 5)
    {
        TuLds& s = reinterpret_cast<TuLds*>(smem)[wv];
        IntraTuLds& ip = reinterpret_cast<IntraTuLds*>(smem + nwvC * sizeof(TuLds))[wv];
        RdoqRef rrC = rdoq_ref_at(smem + (size_t)nwvC * perWave + (size_t)wv * rqBytesC, cLog2);
        rrC.est = s_est[1];
        const uint32_t listed = s_cmode[wv], mode = listed == 36 ? lumaDir : listed;
        for (int b = lane; b < X265AMD_CTX_STRIDE; b += XA_WAVE) s_ctxw[wv][b] = P.ctx[b];
        xa_wave_sync();
        /* U's chain and coefficients, then V's on the contexts U has moved (the flags in front of them live in other contexts: their order against the coefficients
         * is free); the levels are counted where the chain left them, in this wavefront's LDS */
        unsigned long long coeffFrac = 0;
        for (int pl = 0; pl < 2; pl++)
        {
            x265amd_intra_tu_job J = P.ctmpl[pl];
            J.tu.dir_mode = (uint8_t)mode;
            if (cLog2 == 2) { J.tu.fenc = (uint64_t)(uintptr_t)(const void*)s_cfenc[pl]; J.tu.fenc_stride = 4; }
            J.tu.recon += (uint64_t)(2 * wv + pl) * P.slot_pixels * sizeof(pixel);
            J.tu.coeff += (uint64_t)(2 * wv + pl) * P.slot_coeffs * sizeof(int16_t); J.tu.resi += (uint64_t)(2 * wv + pl) * P.slot_coeffs * sizeof(int16_t);
            if (rdoq)
            {
                const RdoqParams rpC = { s_est[1], P.rdoq_lambda2[1 + pl], P.rdoq_lambda[1 + pl], P.psy_rdoq_scale, P.rdoq_level, P.rdoq_tu_depth };
                wave_intra_tu_chain_body<true>(J, nullptr, &s_cres[wv][pl], s, ip, nullptr, lane, nullptr, nullptr, &rrC, &rpC);
            }
            else wave_intra_tu_chain_body<false>(J, nullptr, &s_cres[wv][pl], s, ip, nullptr, lane);
            xa_wave_sync();
            if (s_cres[wv][pl].num_sig) coeffFrac += wave_coeff_bits(s_ctxw[wv], s_ctxw[wv], s.q, cLog2, 1 + pl, 1, (int)mode, P.ctmpl[pl].tu.sign_hide, s_step, lane);
            xa_wave_sync();
        }
        if (lane == 0)
        {
            uint8_t* cw = s_ctxw[wv];
            unsigned long long frac = P.scan_frac;
            frac += cb_bin_t(tabs, cw + 14, listed == 36 ? 0u : 1u);                                /* C_CHROMA_PRED (codeIntraDirChroma, entropy.cpp:1644-1664) */
            if (listed != 36) frac += 2ull << 15;
            /* the two coded block flags share a context (codeSubdivCbfQTChroma at depth 0, both planes: C_QT_CBF + 2), then U's and V's coefficients */
            for (int pl = 0; pl < 2; pl++) frac += cb_bin_t(tabs, cw + CTX_QT_CBF + 2, s_cres[wv][pl].num_sig != 0 ? 1u : 0u);
            frac += coeffFrac;
            unsigned long long dist = 0, energy = 0;
            for (int pl = 0; pl < 2; pl++) { dist += s_cres[wv][pl].nz_dist; energy += s_cres[wv][pl].nz_energy; }
            const unsigned long long bits = (uint32_t)(frac >> 15);
            s_cfrac[wv] = frac;
            s_cost[wv] = P.psy_scale ? dist + ((P.psy_scale * energy) >> 24) + ((bits * P.lambda2) >> 8) : dist + ((bits * P.lambda2 + 128) >> 8);
        }
    }
    __syncthreads();
    if (tid == 0)
    {
        int w = 0;
        unsigned long long best = ~0ull;
        for (int i = 0; i < 5; i++) if (s_cost[i] < best) { best = s_cost[i]; w = i; }
        s_win = w;
        po->chroma_best = (uint32_t)w; po->chroma_reserved = P.pick_sa8d ? s_pickSa8d : 0;
        po->cres[0] = s_cres[w][0]; po->cres[1] = s_cres[w][1];
    }
    __syncthreads();
    {
        const int w = s_win, cn2 = CN * CN;
        int16_t* clOut = P.clevels_dst ? reinterpret_cast<int16_t*>(P.clevels_dst) : &po->clevels[0][0];
        for (int t = tid; t < 2 * cn2; t += nthr)
        {
            const int pl = t >= cn2, i = t - pl * cn2, y = i >> cLog2, x = i & (CN - 1);
            const x265amd_intra_tu_job& C = P.ctmpl[pl];
            const pixel* best = reinterpret_cast<const pixel*>(C.tu.recon) + (size_t)(2 * w + pl) * P.slot_pixels;
            const pixel* last = reinterpret_cast<const pixel*>(C.tu.recon) + (size_t)(2 * 4 + pl) * P.slot_pixels;
            const int16_t* lv = reinterpret_cast<const int16_t*>(C.tu.coeff) + (size_t)(2 * w + pl) * P.slot_coeffs;
            reinterpret_cast<pixel*>(P.crecon_dst[pl])[y * 32 + x] = best[y * C.tu.recon_stride + x];
            if (!P.no_picture) reinterpret_cast<pixel*>(C.nb)[(long)y * C.nb_stride + x] = last[y * C.tu.recon_stride + x];
            clOut[pl * cn2 + i] = lv[i];
        }
    }
    }
    __syncthreads();
    XA_NXN(9);
    /* the other evaluation of a chained CU (one 8x8 unit): its own bits as Search::checkIntra counts them at its end (intra_cu_dev.h's walk), while the deciding
     * workgroup is still busy -- the luma part on the start contexts, the chroma part as the chroma decision left it (chroma's contexts are nobody else's, and a sum
     * of bins does not care for their order) */
    if (P.chain && P.chain_role == 2)
    {
        x265amd_intra_peer* peer = reinterpret_cast<x265amd_intra_peer*>(P.peer);
        const int cwIdx = s_win;
        uint8_t* run = s_ctxw[5];
        if (wv == 0)
        {
            for (int b = lane; b < X265AMD_CTX_STRIDE; b += XA_WAVE) run[b] = P.ctx[b];
            xa_wave_sync();
            const uint32_t mode = s_winMode[0];
            const uint32_t numSigY = s_numSig0;
            unsigned long long frac = P.scan_frac, mvf = 0;
            if (lane == 0)
            {
                frac += cb_bin_t(tabs, run + 8, 1u);                                                   /* C_PART_SIZE: 2Nx2N */
                const int pidx = mode == s_preds0[0] ? 0 : (mode == s_preds0[1] ? 1 : (mode == s_preds0[2] ? 2 : -1));
                frac += cb_bin_t(tabs, run + 13, pidx != -1 ? 1u : 0u);
                frac += (unsigned long long)(pidx != -1 ? 1 + (pidx != 0) : 5) << 15;
                const uint32_t listed = s_cmode[cwIdx];
                mvf = frac + s_enBits[P.ctx[14] ^ (listed == 36 ? 0u : 1u)] + (listed != 36 ? (2ull << 15) : 0ull);
                frac += cb_bin_t(tabs, run + CTX_QT_CBF + 1, numSigY != 0 ? 1u : 0u);
            }
            xa_wave_sync();
            frac = __shfl(frac, 0, 64); mvf = __shfl(mvf, 0, 64);
            if (numSigY) frac += wave_coeff_bits(run, run, P.levels_dst ? reinterpret_cast<const int16_t*>(P.levels_dst) : &po->levels[0][0], unitLog2, 0, 1, (int)mode, P.tmpl[0].tu.sign_hide, s_step, lane);
            xa_wave_sync();
            frac += s_cfrac[cwIdx] - P.scan_frac;
            /* the chroma decision's contexts over the luma walk's: what it moved is chroma's alone */
            for (int b = lane; b < X265AMD_CTX_STRIDE; b += XA_WAVE)
            {
                const uint8_t c = s_ctxw[cwIdx][b];
                peer->fctx[b] = c != P.ctx[b] ? c : run[b];
            }
            if (lane == 0) { peer->ffrac = frac; peer->fmv = mvf; }
        }
        __syncthreads();
        if (tid == 0) xa_chain_publish(&peer->ready, P.chain_token + 1);
    }
    /* The decision of a chained CU (role 1) made in this general form -- what nxn4_decide does for the sixteen-lane form: this CU's bits by the walk of
     * intra_cu_dev.h (Search::checkIntra's count, search.cpp:1254-1275), the other evaluation's record, the two costs, checkBestMode's comparison in the order 2Nx2N,
     * NxN (analysis.cpp:3670-3692), the winner's samples into the picture and the parent's tile, the CU's result for the host, contexts / fraction / modes to the chain.
     * The NxN evaluation's luma is in the picture already (the units predict from each other there); its chroma blocks are in the mode's tile. */
    if (P.chain && P.chain_role == 1 && numUnits == 4 && P.do_chroma)
    {
        __shared__ x265amd_intra_nxn_out s_peerOut;
        __shared__ uint8_t s_fctx[2][X265AMD_CTX_STRIDE];
        __shared__ uint64_t s_ffrac[2], s_fmv[2];
        __shared__ int s_peerOk;
        x265amd_intra_peer* peer = reinterpret_cast<x265amd_intra_peer*>(P.peer);
        x265amd_intra_chain* ch = reinterpret_cast<x265amd_intra_chain*>(P.chain);
        x265amd_intra_cu8_result* out = reinterpret_cast<x265amd_intra_cu8_result*>(P.cu_out);
        const int cwIdx = s_win;
        const uint32_t chromaN = s_cmode[cwIdx];
        const x265amd_intra_tu_job& C0 = P.ctmpl[0];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        XA_CHAIN_START();
        if (wv == 0)
        {
            for (int b = lane; b < X265AMD_CTX_STRIDE; b += XA_WAVE) s_fctx[0][b] = P.ctx[b];
            xa_wave_sync();
            IntraCuBitsIn in;
            in.log2_cu = 3; in.nxn = 1; in.code_part_size = 1; in.inter_slice = 0; in.skip_ctx = 0; in.sign_hide = P.tmpl[0].tu.sign_hide; in.chroma_dir = (uint8_t)chromaN;
            in.cbf_u = s_cres[cwIdx][0].num_sig != 0; in.cbf_v = s_cres[cwIdx][1].num_sig != 0; in.subdiv_flag = 0;
            for (int k = 0; k < 4; k++)
            {
                in.luma_dir[k] = s_winMode[k]; in.cbf_y[k] = s_ures[k].num_sig != 0; in.lev_y[k] = s_lev + 16 * k;
                in.preds[k][0] = s_predsAll[k][0]; in.preds[k][1] = s_predsAll[k][1]; in.preds[k][2] = s_predsAll[k][2];
            }
            in.lev_u = reinterpret_cast<const int16_t*>(P.ctmpl[0].tu.coeff) + (size_t)(2 * cwIdx + 0) * P.slot_coeffs;
            in.lev_v = reinterpret_cast<const int16_t*>(P.ctmpl[1].tu.coeff) + (size_t)(2 * cwIdx + 1) * P.slot_coeffs;
            uint64_t mvf = 0, skipf = 0;
            const uint64_t frac = wave_intra_cu_bits(in, s_fctx[0], P.scan_frac, &mvf, &skipf, s_step, tabs, lane);
            if (lane == 0) { s_ffrac[0] = frac; s_fmv[0] = mvf; }
        }
        if (tid == 64)
        {
            bool ok = xa_chain_wait(&peer->ready, P.chain_token + 1);
            if (ok && __hip_atomic_load(&peer->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ~0ull) ok = false;      /* the other side gave up */
            s_peerOk = ok ? 1 : 0;
        }
        __syncthreads();
        XA_CHAIN(2);
        if (!s_peerOk)
        {
            if (tid == 0) { out->status = 2; xa_chain_publish(&ch->seq, P.chain_token + 1); }
            return;
        }
        for (int i = tid; i < (int)(sizeof(x265amd_intra_nxn_out) / 8); i += nthr) reinterpret_cast<uint64_t*>(&s_peerOut)[i] = reinterpret_cast<const uint64_t*>(&peer->out)[i];
        for (int i = tid; i < X265AMD_CTX_STRIDE; i += nthr) s_fctx[1][i] = peer->fctx[i];
        if (tid == 0) { s_ffrac[1] = peer->ffrac; s_fmv[1] = peer->fmv; }
        __syncthreads();
        XA_CHAIN(3);
        const x265amd_intra_nxn_out& Q = s_peerOut;
        const uint32_t chroma2 = nxn4_chroma_stored(Q.mode[0], Q.chroma_best);
        const uint32_t lumaN = (uint32_t)(s_ures[0].nz_dist + s_ures[1].nz_dist + s_ures[2].nz_dist + s_ures[3].nz_dist);
        const uint32_t chromaDN = (uint32_t)(s_cres[cwIdx][0].nz_dist + s_cres[cwIdx][1].nz_dist), chromaD2 = (uint32_t)(Q.cres[0].nz_dist + Q.cres[1].nz_dist);
        const uint32_t luma2 = (uint32_t)Q.res[0].nz_dist;
        const uint32_t psyN = P.psy_scale ? s_psyNxn : 0u, psy2 = P.psy_scale ? Q.res[0].nz_energy : 0u;
        const uint32_t bitsN = (uint32_t)(s_ffrac[0] >> 15), bits2 = (uint32_t)(s_ffrac[1] >> 15);
        const uint64_t distN = (uint64_t)lumaN + chromaDN, dist2 = (uint64_t)luma2 + chromaD2;
        const uint64_t costN = P.psy_scale ? distN + ((P.psy_scale * (uint64_t)psyN) >> 24) + (((uint64_t)bitsN * P.lambda2) >> 8) : distN + (((uint64_t)bitsN * P.lambda2 + 128) >> 8);
        const uint64_t cost2 = P.psy_scale ? dist2 + ((P.psy_scale * (uint64_t)psy2) >> 24) + (((uint64_t)bits2 * P.lambda2) >> 8) : dist2 + (((uint64_t)bits2 * P.lambda2 + 128) >> 8);
        const bool nxnWins = costN < cost2;
        const int win = nxnWins ? 0 : 1;
        {
            const x265amd_intra_tu_job& T0 = P.tmpl[0];
            pixel* pic = reinterpret_cast<pixel*>(T0.nb);
            pixel* dstY = reinterpret_cast<pixel*>(P.win_dst[0]);
            const pixel* peerY = reinterpret_cast<const pixel*>(P.peer_recon[0]);
            if (tid < 64)
            {
                const int y = tid >> 3, x = tid & 7;
                const pixel v = nxnWins ? pic[(long)y * T0.nb_stride + x] : peerY[y * 64 + x];
                if (!nxnWins) pic[(long)y * T0.nb_stride + x] = v;
                dstY[y * 64 + x] = v;
            }
            else if (tid < 96)
            {
                const int pl = (tid - 64) >> 4, i = tid & 15, y = i >> 2, x = i & 3;
                const x265amd_intra_tu_job& C = P.ctmpl[pl];
                const pixel v = nxnWins ? reinterpret_cast<const pixel*>(P.crecon_dst[pl])[y * 32 + x] : reinterpret_cast<const pixel*>(P.peer_recon[1 + pl])[y * 32 + x];
                reinterpret_cast<pixel*>(C.nb)[(long)y * C.nb_stride + x] = v;
                reinterpret_cast<pixel*>(P.win_dst[1 + pl])[y * 32 + x] = v;
            }
            else if (tid < 96 + 96)
            {
                const int i = tid - 96;
                int16_t v;
                if (i < 64) v = nxnWins ? s_lev[i] : Q.levels[0][i];
                else
                {
                    const int pl = (i - 64) >> 4;
                    v = nxnWins ? (reinterpret_cast<const int16_t*>(P.ctmpl[pl].tu.coeff) + (size_t)(2 * cwIdx + pl) * P.slot_coeffs)[i & 15] : Q.clevels[pl][i & 15];
                }
                out->levels[i] = v;
            }
            else if (tid < 192 + X265AMD_CTX_STRIDE)
            {
                const int i = tid - 192;
                const uint8_t v = s_fctx[win][i];
                out->ctx[i] = v; ch->ctx[i] = v;
            }
        }
        (void)C0;
        if (tid == 0)
        {
            out->rd_cost = nxnWins ? costN : cost2; out->other_cost = nxnWins ? cost2 : costN; out->frac_bits = s_ffrac[win];
            const uint32_t tb = nxnWins ? bitsN : bits2, mvb = (uint32_t)(s_fmv[win] >> 15);
            out->total_bits = tb; out->mv_bits = mvb; out->coeff_bits = tb - mvb;
            out->psy_energy = nxnWins ? psyN : psy2; out->res_energy = nxnWins ? s_resNxn : (uint32_t)Q.res[0].zero_dist;
            out->luma_dist = nxnWins ? lumaN : luma2; out->chroma_dist = nxnWins ? chromaDN : chromaD2;
            out->part_size = nxnWins ? 3 : 0; out->chroma_dir = (uint8_t)(nxnWins ? chromaN : chroma2);
            out->cbf_u = nxnWins ? (s_cres[cwIdx][0].num_sig != 0) : (Q.cres[0].num_sig != 0); out->cbf_v = nxnWins ? (s_cres[cwIdx][1].num_sig != 0) : (Q.cres[1].num_sig != 0);
            for (int k = 0; k < 4; k++)
            {
                const uint8_t m = nxnWins ? s_winMode[k] : Q.mode[0];
                out->luma_dir[k] = m; ch->mode[P.chain_index & 3][k] = m;
                out->cbf_y[k] = nxnWins ? (s_ures[k].num_sig != 0) : (Q.res[0].num_sig != 0);
            }
            ch->frac = s_ffrac[win];
            out->status = 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        XA_CHAIN(4);
        if (tid == 0) xa_chain_publish(&ch->seq, P.chain_token + 1);
        XA_CHAIN(5);
    }
}

#endif
