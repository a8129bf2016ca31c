/* The 8x8 CU coded NxN -- four 4x4 luma units and one 4x4 block per chroma plane -- with everything between the command's first and last instruction in LDS and
 * registers (include/x265amd.h: x265amd_intra_nxn with num_units 4; the same results as block_intra_nxn in intra_pu_dev.h, which stays the form of every other case).
 *
 * A 4x4 transform chain is sixteen samples: on a wavefront of its own fifteen sixteenths of the lanes idle and every step waits for an LDS or memory round trip, which is
 * where the 62 microseconds of the general form go (profiles/r03_queue_profile_*.txt: 7 us per unit in the chains and their bits, 2.8 us in the scan, 1 us each in the
 * neighbour gather from the picture and the winner's copy back to it).  Here
 *  - the CU's neighbourhood is read from the picture once into a 17x17 frame in LDS; a unit's reconstruction goes into the frame and the next unit gathers its
 *    neighbours there (Search::estIntraPredQT's initAdiPattern per unit, search.cpp:1509-1696, predict.cpp:579-700); the picture, the tiles and the levels are
 *    written once at the end;
 *  - a candidate is a group of sixteen lanes (one per sample, four candidates per wavefront): prediction sample, residual, the two passes of the 4x4 DST / DCT and their
 *    inverses as cross-lane gathers (dct.cpp:83-110, :440-520), quantisation with sign-bit hiding (dct.cpp:664-686, quant.cpp:247-395), reconstruction, both
 *    distortions and psy energies -- no LDS traffic except the neighbour arrays and two small tables;
 *  - its bits: the context walks of wave_coeff_bits_4x4 (entropy_dev.h) folded onto the sixteen lanes of the group.
 * Integer arithmetic throughout, the same operations in the same order as the general form: the results are identical (tests/test_intra_nxn.py). */
#ifndef X265AMD_INTRA_NXN4_DEV_H
#define X265AMD_INTRA_NXN4_DEV_H
#include "tu_dev.h"
#include "intra_dev.h"
#include "entropy_dev.h"
#include "intra_cu_dev.h"

struct Nxn4Tabs
{
    int8_t angle[17]; int16_t invAngle[8];
    int16_t T[2][16];               /* [0] DCT 4x4, [1] DST 4x4 */
    uint8_t scan[3][16], inv[3][16];    /* raster position of scan position k; scan position of raster position r */
    uint16_t sigMask[3][9];         /* scan positions whose significance context is c (4x4 map) */
};

XA_DEV void nxn4_fill_tabs(Nxn4Tabs& tb, int tid)
{
    if (tid < 17) tb.angle[tid] = xa_tbl.angle[tid];
    if (tid < 8) tb.invAngle[tid] = xa_tbl.invAngle[tid];
    if (tid < 16) { tb.T[0][tid] = xa_tbl.dct[0][tid]; tb.T[1][tid] = xa_tbl.dst4[tid]; }
    if (tid < 48)
    {
        const int t = tid >> 4, k = tid & 15;
        const uint32_t rr = cb_in_cg(t, k);
        tb.scan[t][k] = (uint8_t)rr; tb.inv[t][rr] = (uint8_t)k;
    }
    if (tid >= 64 && tid < 64 + 27)
    {
        const int t = (tid - 64) / 9, c = (tid - 64) % 9;
        uint32_t m = 0;
        for (int k = 0; k < 16; k++) if (cb_sig_ctx_inc(2, 0, cb_in_cg(t, k)) == (uint32_t)c) m |= 1u << k;
        tb.sigMask[t][c] = (uint16_t)m;
    }
}

/* one prediction sample of a 4x4 block (intrapred.cpp:54-209): ref = [0] above-left, [1..8] above + above-right, [9..16] left + below-left; sw = the mirrored copy the
 * horizontal modes read; edge: the DC / pure vertical / pure horizontal edge filters (luma only) */
XA_DEV int nxn4_pred_sample(const pixel* ref, const pixel* sw, const Nxn4Tabs& tb, int mode, int dc, int y, int x, bool edge)
{
    if (mode == 0)
    {
        const pixel* above = ref + 1; const pixel* left = ref + 9;
        return ((3 - x) * left[y] + (3 - y) * above[x] + (x + 1) * above[4] + (y + 1) * left[4] + 4) >> 3;
    }
    if (mode == 1)
    {
        const pixel* above = ref + 1; const pixel* left = ref + 9;
        if (edge)
        {
            if (x == 0 && y == 0) return (above[0] + left[0] + 2 * dc + 2) >> 2;
            if (y == 0) return (above[x] + 3 * dc + 2) >> 2;
            if (x == 0) return (left[y] + 3 * dc + 2) >> 2;
        }
        return dc;
    }
    const bool hor = mode < 18;
    const int angOff = hor ? 10 - mode : mode - 26;
    const int angle = tb.angle[8 + angOff];
    const int invAngle = angle < 0 ? tb.invAngle[-angOff - 1] : 0;
    return hor ? ang_sample(sw, 4, angle, invAngle, edge, x, y) : ang_sample(ref, 4, angle, invAngle, edge, y, x);
}

/* a unit's 17 neighbour samples from the frame (fillReferenceSamples with the unit's availability flags: bit 0 below-left, 1 left, 2 above-left, 3 above, 4 above-right),
 * the mirrored copy and the DC value: lanes 0..16 of the calling wavefront; org: the unit's first sample in the frame, stride fs */
XA_DEV int nxn4_neighbours(const pixel* org, int fs, uint32_t avail, pixel* ref, pixel* sw, int lane)
{
    avail &= 31u;
    if (lane <= 16)
    {
        const int i = lane;                                         /* substitution order: 0 = bottom-most below-left ... 8 = above-left ... 16 */
        const int u = i < 8 ? i >> 2 : (i == 8 ? 2 : 3 + ((i - 9) >> 2));
        int src = i;
        if (!((avail >> u) & 1))
        {
            const uint32_t before = avail & ((1u << u) - 1);
            if (before) { const int b = 31 - __clz((int)before); src = b < 2 ? 4 * b + 3 : (b == 2 ? 8 : 9 + 4 * (b - 3) + 3); }
            else if (avail) { const int a = __ffs((int)avail) - 1; src = a < 2 ? 4 * a : (a == 2 ? 8 : 9 + 4 * (a - 3)); }
            else src = -1;
        }
        int v;
        if (src < 0) v = 1 << (XA_DEPTH - 1);
        else if (src < 8) v = org[(7 - src) * fs - 1];
        else if (src == 8) v = org[-fs - 1];
        else v = org[-fs + (src - 9)];
        const int d = i == 8 ? 0 : (i > 8 ? i - 8 : 8 + (8 - i));
        ref[d] = (pixel)v;
        /* the mirrored copy: above <-> left */
        sw[d == 0 ? 0 : (d <= 8 ? d + 8 : d - 8)] = (pixel)v;
    }
    xa_wave_sync();
    const int part = lane < 4 ? (int)ref[1 + lane] + (int)ref[9 + lane] : 0;
    int s = part;
    s += __builtin_amdgcn_update_dpp(0, s, 0xB1, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x4E, 0xf, 0xf, true);
    return (__builtin_amdgcn_readfirstlane(s) + 4) >> 3;
}

struct Q4 { int qc, qbits, add, scale, shiftq; };
XA_DEV Q4 q4_make(int qpScaled, int sliceType)
{
    Q4 q;
    const int rem = qpScaled % 6, per = qpScaled / 6;
    const int transformShift = 15 - XA_DEPTH - 2;
    q.qc = tu_quantScales[rem]; q.qbits = 14 + per + transformShift; q.add = (sliceType == 2 ? 171 : 85) << (q.qbits - 9);
    q.scale = tu_invQuantScales[rem] << per; q.shiftq = 20 - 14 - transformShift;
    return q;
}

struct Chain4 { int lv, pred, rec; uint32_t numSig, nzDist, nzEnergy, zeroDist, zeroEnergy; };

/* RDOQ (round 5): a group's working set for wave_rdo_quant's sixteen-lane form (tu_dev.h) -- the coefficients, the levels and the per-position records of ONE 4x4 block;
 * the bit-estimate table is the command's (Nxn4Lds::est), read where it lies */
struct Rq4Area { Tu16 t; int64_t costSig[16], delta[16], costCg[2]; int32_t rateDown[16], sigDelta[16]; };
static_assert(sizeof(Rq4Area) == 560, "");
/* what a candidate's chain needs to quantise that way: a: the group's area (null: plain quantisation); fD: the source block's DCT coefficient at this lane's raster
 * position (psy-rdoq: copy_ps + dct, quant.cpp:436-440) */
struct Rq4 { Rq4Area* a; const RdoqParams* P; int ttype, dirMode, qpScaled, fD; };

/* the two forward passes of a 4x4 block held one sample per lane (grp_fwd_pass: out[k][j] = sum_n T[k][n] in[j][n]) */
XA_DEV int grp16_forward4(int r, const int16_t* T, int lane)
{
    const int l = lane & 15, base = lane & 48, hi = l >> 2, lo = l & 3;
    const int tf0 = T[hi * 4], tf1 = T[hi * 4 + 1], tf2 = T[hi * 4 + 2], tf3 = T[hi * 4 + 3];
    const int g0 = base + lo * 4;
    const int s1 = 1 + XA_DEPTH - 8;
    int a0 = __shfl(r, g0, 64), a1 = __shfl(r, g0 + 1, 64), a2 = __shfl(r, g0 + 2, 64), a3 = __shfl(r, g0 + 3, 64);
    const int b = (int)(int16_t)((tf0 * a0 + tf1 * a1 + tf2 * a2 + tf3 * a3 + (1 << (s1 - 1))) >> s1);
    a0 = __shfl(b, g0, 64); a1 = __shfl(b, g0 + 1, 64); a2 = __shfl(b, g0 + 2, 64); a3 = __shfl(b, g0 + 3, 64);
    return (int)(int16_t)((tf0 * a0 + tf1 * a1 + tf2 * a2 + tf3 * a3 + 128) >> 8);
}

/* the energy term of psyCost_pp for a 4x4 block held one sample per lane (pixel.cpp:744-775): satd against zeros minus a quarter of the sum */
XA_DEV int nxn4_energy(int v, int lane)
{
    const int h = xa_lane_had4x4(v, lane);
    const int sum = xa_row16_sum(abs(h));
    const int dcv = __shfl(h, lane & 48, 64);
    return (sum >> 1) - (dcv >> 2);
}

/* the transform chain of one 4x4 candidate on the sixteen lanes of a group: f / p = this lane's source and prediction sample (raster order).  Returns with .lv = the
 * level at this lane's RASTER position; numSig and the measurements are the same in all sixteen lanes. */
template<bool RQ = false>
XA_DEV Chain4 grp16_chain4(int f, int p, int dst, const Q4& q, int signHide, int scanType, const Nxn4Tabs& tb, int srcEnergy, int lane, const Rq4* rq = nullptr)
{
    const int l = lane & 15, base = lane & 48, hi = l >> 2, lo = l & 3;
    const int16_t* T = tb.T[dst];
    const int ti0 = T[lo], ti1 = T[4 + lo], ti2 = T[8 + lo], ti3 = T[12 + lo];
    const int r = f - p;
    const int c = grp16_forward4(r, T, lane);
    int a0, a1, a2, a3;
    Chain4 o;
    uint32_t numSig;
    if constexpr (RQ)
    {
        /* Quant::rdoQuant on the group's sixteen lanes: coefficients and levels by raster position in the group's area */
        Rq4Area& A = *rq->a;
        const bool usePsy = rq->P->psyRdoqScale != 0 && rq->ttype == 0;
        A.t.dct[l] = (int16_t)c;
        if (usePsy) reinterpret_cast<int16_t*>(A.t.deltaU)[l] = (int16_t)rq->fD;
        xa_wave_sync();
        RdoqRef rr{ A.costSig, A.delta, A.rateDown, A.sigDelta, A.costCg, const_cast<int32_t*>(rq->P->est) };
        numSig = wave_rdo_quant<RdoqRef, true, Tu16>(A.t, rr, *rq->P, 2, rq->ttype, 1, rq->dirMode, rq->qpScaled, signHide, usePsy, lane);
        xa_wave_sync();
        o.lv = A.t.q[l];
    }
    else
    {
    /* from here to the levels: lane l = scan position l */
    const int cs = __shfl(c, base + tb.scan[scanType][l], 64);
    const int sign = cs < 0 ? -1 : 1;
    const int tmplevel = abs(cs) * q.qc;
    const int level = (tmplevel + q.add) >> q.qbits;
    const int dU = (tmplevel - (level << q.qbits)) >> (q.qbits - 8);
    int lv = xa_clip3(-32768, 32767, level * sign);
    const uint32_t nz = (uint32_t)(__ballot(lv != 0) >> base) & 0xFFFFu;
    numSig = (uint32_t)__popc(nz);
    {
        /* signBitHidingHDQ for the one coefficient group (quant.cpp:247-395), as grp_tu_forward has it */
        const int firstNZ = nz ? __builtin_ctz(nz) : -1, lastNZ = nz ? 31 - __builtin_clz(nz) : -1;
        const int absSum = xa_row16_sum(lv);
        const int cFirst = __shfl(lv, base + (firstNZ < 0 ? 0 : firstNZ), 64);
        const uint32_t signbit = cFirst > 0 ? 0 : 1;
        const bool active = signHide && numSig >= 2 && lastNZ - firstNZ >= 4 && signbit != ((uint32_t)absSum & 1);
        int curCost, curChange = 1;
        if (lv)
        {
            if (dU > 0) curCost = -dU;
            else if (l == firstNZ && abs(lv) == 1) curCost = 0x7fffffff;
            else { curCost = dU; curChange = -1; }
        }
        else if (l < firstNZ) curCost = ((cs >= 0 ? 0u : 1u) != signbit) ? 0x7fffffff : -dU;
        else curCost = -dU;
        long long key = l <= lastNZ ? (((long long)curCost) << 4) | (long long)(15 - l) : 0x7fffffffffffffffLL;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { const long long other = __shfl_xor(key, o, 64); key = other < key ? other : key; }
        int delta = 0;
        if (active && (int)(key & 15) == 15 - l)
        {
            int finalChange = curChange;
            if (lv == 32767 || lv == -32768) finalChange = -1;
            if (!lv) delta = 1;
            else if (finalChange == -1 && abs(lv) == 1) delta = -1;
            const int sigMask = cs >> 31;
            lv = (int)(int16_t)(lv + ((finalChange ^ sigMask) - sigMask));
        }
        numSig = (uint32_t)((int)numSig + xa_row16_sum(delta));
    }
    /* back to raster order */
    o.lv = __shfl(lv, base + tb.inv[scanType][l], 64);
    }
    int resi = 0;
    if (numSig)
    {
        const int dq = xa_clip3(-32768, 32767, (o.lv * q.scale + (1 << (q.shiftq - 1))) >> q.shiftq);       /* dequant_normal_c (dct.cpp:612-634) */
        const int lv0 = __shfl(o.lv, base, 64), dq0 = __shfl(dq, base, 64);
        if (numSig == 1 && lv0 != 0 && !dst)
        {
            /* DC only (quant.cpp:588-598) */
            const int shift2 = 12 - (XA_DEPTH - 8) - 3;
            resi = (int)(int16_t)(((((dq0 + 1) >> 1) * 8) + (1 << (shift2 - 1))) >> shift2);
        }
        else
        {
            /* inverse passes (grp_inv_pass: out[j][n] = sum_k T[k][n] in[k][j]) */
            const int h0 = base + hi;
            a0 = __shfl(dq, h0, 64); a1 = __shfl(dq, h0 + 4, 64); a2 = __shfl(dq, h0 + 8, 64); a3 = __shfl(dq, h0 + 12, 64);
            const int t = xa_clip3(-32768, 32767, (ti0 * a0 + ti1 * a1 + ti2 * a2 + ti3 * a3 + 64) >> 7);
            a0 = __shfl(t, h0, 64); a1 = __shfl(t, h0 + 4, 64); a2 = __shfl(t, h0 + 8, 64); a3 = __shfl(t, h0 + 12, 64);
            const int s2 = 12 - (XA_DEPTH - 8);
            resi = xa_clip3(-32768, 32767, (ti0 * a0 + ti1 * a1 + ti2 * a2 + ti3 * a3 + (1 << (s2 - 1))) >> s2);
        }
    }
    o.pred = p;
    o.rec = xa_clip_pixel(p + resi);
    o.numSig = numSig;
    const int e = f - o.rec;
    o.zeroDist = (uint32_t)xa_row16_sum(r * r);
    o.zeroEnergy = (uint32_t)abs(srcEnergy - nxn4_energy(p, lane));
    if (numSig)
    {
        o.nzDist = (uint32_t)xa_row16_sum(e * e);
        o.nzEnergy = (uint32_t)abs(srcEnergy - nxn4_energy(o.rec, lane));
    }
    else { o.nzDist = o.zeroDist; o.nzEnergy = o.zeroEnergy; }
    return o;
}

/* bits-only codeCoeffNxN of a 4x4 unit on the sixteen lanes of a group: lvScan = the level at scan position (lane & 15).  The context walks of wave_coeff_bits_4x4 with
 * the owners folded onto sixteen lanes: lanes 0..8 the significance contexts, 9..12 the greater-1 contexts, 13 the greater-2 context; then lanes 0..5 the bins of the
 * last position.  ctx: the start states (LDS); ctxOut (LDS, may be ctx, may be null): the states behind the unit.  All sixteen lanes return the FIX15 total. */
XA_DEV uint32_t grp16_coeff_bits4(const uint8_t* ctx, uint8_t* ctxOut, int lvScan, int isLuma, int scanType, int signHide, const uint32_t* step, const Nxn4Tabs& tb, int lane)
{
    const int l = lane & 15, base = lane & 48;
    const uint32_t a = (uint32_t)(lvScan < 0 ? -lvScan : lvScan);
    const uint32_t sigM = (uint32_t)(__ballot(a != 0) >> base) & 0xFFFFu;
    const uint32_t gt1M = (uint32_t)(__ballot(a > 1) >> base) & 0xFFFFu, gt2M = (uint32_t)(__ballot(a > 2) >> base) & 0xFFFFu;
    if (!sigM) return 0;
    const int lastK = 31 - __clz((int)sigM), firstK = __ffs((int)sigM) - 1;
    const uint32_t nnz = (uint32_t)__popc(sigM);
    uint32_t first8 = sigM;
    for (uint32_t drop = nnz > 8 ? nnz - 8 : 0; drop; drop--) first8 &= first8 - 1;
    const uint32_t g1 = gt1M & first8;
    const int s1K = g1 ? 31 - __clz((int)g1) : -1;
    uint32_t sum = 0;
    {
        uint32_t todo = 0, flags = 0;
        int ci = 0;
        if (l < 9)
        {
            todo = (uint32_t)tb.sigMask[scanType][l] & ((1u << lastK) - 1u); flags = sigM;
            ci = CTX_SIG + (isLuma ? 0 : N_SIG_LUMA) + l;
        }
        else if (l < 13)
        {
            const int j = l - 9;
            const uint32_t pos0 = 1u << lastK;
            const uint32_t rem = first8 & ~pos0;
            const uint32_t pos1 = rem ? 1u << (31 - __clz((int)rem)) : 0u;
            const uint32_t rem2 = rem & ~pos1;
            const uint32_t below = s1K >= 0 ? first8 & ((1u << s1K) - 1u) : 0u;
            todo = j == 0 ? below : (j == 1 ? pos0 : (j == 2 ? pos1 & ~below : rem2 & ~below));
            flags = gt1M;
            ci = CTX_ONE + (isLuma ? 0 : N_ONE_LUMA) + j;
        }
        else if (l == 13)
        {
            todo = s1K >= 0 ? 1u << s1K : 0u; flags = gt2M;
            ci = CTX_ABS + (isLuma ? 0 : N_ABS_LUMA);
        }
        if (todo)
        {
            uint32_t st = ctx[ci];
            do
            {
                const int k = 31 - __clz((int)todo);
                todo &= ~(1u << k);
                const uint32_t e = step[(st << 1) | ((flags >> k) & 1u)];
                sum += e & 0xFFFFFFu; st = e >> 24;
            } while (todo);
            if (ctxOut) ctxOut[ci] = (uint8_t)st;
        }
    }
    if (l < 6)
    {
        /* last position: bin i of the x (lanes 0..2) or y (3..5) prefix, each in a context of its own */
        const uint32_t rr = tb.scan[scanType][lastK];
        uint32_t px = rr & 3, py = rr >> 2;
        if (scanType == 2) { const uint32_t t = px; px = py; py = t; }
        const int isY = l >= 3, i = l - (isY ? 3 : 0);
        const uint32_t pos = isY ? py : px;
        const bool one = (uint32_t)i < pos, zero = (uint32_t)i == pos && pos < 3;
        if (one || zero)
        {
            const int ci = CTX_LAST_X + (isLuma ? 0 : N_LAST_XY_LUMA) + (isY ? N_LAST_XY : 0) + i;
            const uint32_t e = step[((uint32_t)ctx[ci] << 1) | (one ? 1u : 0u)];
            sum += e & 0xFFFFFFu;
            if (ctxOut) ctxOut[ci] = (uint8_t)(e >> 24);
        }
    }
    /* sign bits and escape codes (costCoeffRemain_c): serial in the Rice parameter, plain arithmetic on values fetched from the group */
    uint32_t bypass = nnz - ((signHide && lastK - firstK >= 4) ? 1u : 0u);
    const uint32_t startIdx = s1K >= 0 ? (uint32_t)__popc(sigM >> (s1K + 1)) : 8u;
    if (nnz > startIdx)
    {
        uint32_t rest = s1K >= 0 ? sigM & ((2u << s1K) - 1u) : sigM & ~first8;
        uint32_t idx = startIdx, rice = 0;
        int baseLevel = 3;
        while (rest)
        {
            const int k = 31 - __clz((int)rest);
            rest &= ~(1u << k);
            if (idx >= 8) baseLevel = 1;
            const uint32_t av = (uint32_t)__shfl((int)a, base + k, 64);
            int code = (int)av - baseLevel;
            if (code >= 0)
            {
                code = (int)((uint32_t)code >> rice) - 3;
                if (code >= 0) { const uint32_t length = 31 - (uint32_t)__clz(code + 1); code = (int)(length + length); }
                bypass += (uint32_t)(3 + 1 + (int)rice + code);
                if (av > (3u << rice)) rice = (rice + 1) - (rice >> 2);
            }
            baseLevel = 2;
            idx++;
        }
    }
    return (uint32_t)xa_row16_sum((int)sum) + (bypass << 15);
}

/* ---- the command ---- */
struct Nxn4Lds
{
    Nxn4Tabs tb;
    pixel frame[17 * 17];           /* frame[(y + 1) * 17 + x + 1]: the CU's 8x8 samples and what lies above / left of them (x, y = -1 .. 15) */
    pixel fenc[64], pred[64];
    int16_t lev[64];
    pixel ref[8][20], sw[8][20];    /* per wavefront: the current unit's neighbours */
    int32_t sa8d[36];
    unsigned long long cost[16];
    x265amd_tu_result res[16];
    uint8_t winMode[4];
    /* chroma */
    pixel cref[2][20], csw[2][20], cfenc[2][16], crec[6][2][16];     /* (six: the five listed modes and, when they are evaluated ahead of the luma decision, a derived mode outside them) */
    int16_t clev[6][2][16];
    x265amd_tu_result cres[6][2];
    uint8_t ctxw[6][X265AMD_CTX_STRIDE];
    uint32_t step[256], enBits[128];
    uint8_t enLps[64];
    /* the CU's own results, kept for the decision of a chained CU */
    x265amd_tu_result ures[4];
    uint8_t preds[4][4];
    uint32_t psyNxn, resNxn, cw;
    /* a chained CU: the other evaluation's record, the two final context sets, their fractions */
    x265amd_intra_nxn_out peerOut;
    uint8_t fctx[2][X265AMD_CTX_STRIDE];
    uint64_t ffrac[2], fmv[2];
    int peerOk;
    /* the chained CU's own bits, counted beside the evaluation by the last wavefront: the luma part's running contexts and fraction, the fraction behind the luma
     * prediction info; the chroma modes' fractions as the chroma decision counted them */
    uint8_t runCtx[X265AMD_CTX_STRIDE];
    uint64_t runFrac, runMv;
    uint64_t cfrac[6], ccoef[6];
    unsigned long long ccost[5];                 /* the chroma modes' costs (apart from the luma candidates': the two decisions overlap) */
    /* RDOQ: the command's two bit-estimate tables (luma units, chroma blocks: Entropy::estBit on its start contexts), and where the groups' areas lie (32 of them) */
    int32_t est[2][184];
    Rq4Area* rq;
};
/* what a command with RDOQ needs behind its Nxn4Lds */
#define NXN4_RQ_BYTES (32 * (int)sizeof(Rq4Area))

/* estIntraPredChromaQT for the one 4x4 block per plane of an 8x8 CU (search.cpp:1754-1889): the five listed modes, a group of sixteen lanes each (cwv 0: modes 0..3,
 * cwv 1: mode 4), one plane per call -- U (pl 0: from a copy of the start contexts), then V (pl 1: on the contexts U has moved, and the mode's cost).  Reads S.cref /
 * S.csw / S.cfenc and the tables; leaves per mode S.crec, S.clev, S.cres, S.ctxw (the contexts behind the mode's bins), S.cfrac (the coder's fraction behind them, from
 * P.scan_frac) and S.ccost.  Two calls with a wavefront-level fence between them; the caller synchronises the workgroup and picks. */
XA_DEV void nxn4_chroma_plane(const x265amd_intra_nxn_job& P, Nxn4Lds& S, const uint32_t list[5], uint32_t lumaDir, int pl, const EnTabs& tabs, int lane, int cwv, int grp, int l, Rq4Area* area = nullptr)
{
    const bool rdoq = P.rdoq_level != 0;      /* (then area: the calling group's) */
    const int mi = cwv * 4 + grp, m = mi < 5 ? mi : 4;
    const uint32_t listed = list[m], mode = listed == 36 ? lumaDir : listed;
    const int scanType = mode >= 22 && mode <= 30 ? 1 : (mode >= 6 && mode <= 14 ? 2 : 0);
    const int y = l >> 2, x = l & 3;
    uint8_t* cw = S.ctxw[m];
    if (pl == 0)
    {
        if (mi < 5) for (int b = l; b < X265AMD_CTX_STRIDE; b += 16) cw[b] = P.ctx[b];
        if (mi < 5 && l == 0) S.ccoef[m] = 0;
        xa_wave_sync();
    }
    const x265amd_intra_tu_job& C = P.ctmpl[pl];
    const Q4 qC = q4_make(C.tu.qp_scaled, C.tu.slice_type);
    const pixel* cr = S.cref[pl];
    const int part = l < 4 ? (int)cr[1 + l] + (int)cr[9 + l] : 0;
    int s = part;
    s += __builtin_amdgcn_update_dpp(0, s, 0xB1, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0, s, 0x4E, 0xf, 0xf, true);
    const int dc = (__shfl(s, lane & 48, 64) + 4) >> 3;
    const int f = S.cfenc[pl][l];
    const int p = nxn4_pred_sample(cr, S.csw[pl], S.tb, (int)mode, dc, y, x, false);
    Chain4 ch;
    if (rdoq)
    {
        const RdoqParams rp = { S.est[1], P.rdoq_lambda2[1 + pl], P.rdoq_lambda[1 + pl], P.psy_rdoq_scale, P.rdoq_level, P.rdoq_tu_depth };
        const Rq4 rq = { area, &rp, 1 + pl, (int)mode, C.tu.qp_scaled, 0 };
        ch = grp16_chain4<true>(f, p, 0, qC, C.tu.sign_hide, scanType, S.tb, nxn4_energy(f, lane), lane, &rq);
    }
    else ch = grp16_chain4(f, p, 0, qC, C.tu.sign_hide, scanType, S.tb, nxn4_energy(f, lane), lane);
    const int lvScan = __shfl(ch.lv, (lane & 48) + S.tb.scan[scanType][l], 64);
    if (mi < 5)
    {
        const uint32_t cf = ch.numSig ? grp16_coeff_bits4(cw, cw, lvScan, 0, scanType, C.tu.sign_hide, S.step, S.tb, lane) : 0u;
        S.crec[m][pl][l] = (pixel)ch.rec; S.clev[m][pl][l] = (int16_t)ch.lv;
        if (l == 0)
        {
            x265amd_tu_result r;
            r.num_sig = ch.numSig; r.zero_energy = ch.zeroEnergy; r.nz_energy = ch.nzEnergy; r.reserved = 0; r.zero_dist = ch.zeroDist; r.nz_dist = ch.nzDist;
            S.cres[m][pl] = r;
            S.ccoef[m] += cf;
        }
    }
    xa_wave_sync();
    if (pl == 1 && mi < 5 && l == 0)
    {
        const x265amd_tu_result rU = S.cres[m][0], rV = S.cres[m][1];
        unsigned long long frac = P.scan_frac;
        frac += cb_bin_t(tabs, cw + 14, listed == 36 ? 0u : 1u);                                /* C_CHROMA_PRED (codeIntraDirChroma, entropy.cpp:1644-1664) */
        if (listed != 36) frac += 2ull << 15;
        frac += cb_bin_t(tabs, cw + CTX_QT_CBF + 2, rU.num_sig != 0 ? 1u : 0u);                 /* the two coded block flags share a context */
        frac += cb_bin_t(tabs, cw + CTX_QT_CBF + 2, rV.num_sig != 0 ? 1u : 0u);
        frac += S.ccoef[m];
        const unsigned long long dist = rU.nz_dist + rV.nz_dist, energy = (unsigned long long)rU.nz_energy + rV.nz_energy;
        const unsigned long long bits = (uint32_t)(frac >> 15);
        S.cfrac[m] = frac;
        S.ccost[m] = P.psy_scale ? dist + ((P.psy_scale * energy) >> 24) + ((bits * P.lambda2) >> 8) : dist + ((bits * P.lambda2 + 128) >> 8);
    }
}
/* both planes by wavefronts 0 and 1 of the caller */
XA_DEV void nxn4_chroma_modes(const x265amd_intra_nxn_job& P, Nxn4Lds& S, const uint32_t list[5], uint32_t lumaDir, const EnTabs& tabs, int lane, int wv, int grp, int l)
{
    if (wv >= 2) return;
    Rq4Area* area = P.rdoq_level ? S.rq + wv * 4 + grp : nullptr;
    nxn4_chroma_plane(P, S, list, lumaDir, 0, tabs, lane, wv, grp, l, area);
    xa_wave_sync();
    nxn4_chroma_plane(P, S, list, lumaDir, 1, tabs, lane, wv, grp, l, area);
}

/* The same evaluation for chroma modes given by NUMBER, ahead of the luma decision (block_intra_nxn, an 8x8 CU coded 2Nx2N): slot = slot0 + the lane's group, active when
 * slot < slotEnd; modes[slot] is the prediction mode.  What depends on how the mode will be SIGNALLED -- the bin of intra_chroma_pred_mode and its context -- is left out:
 * S.cfrac[slot] is the coded block flags' and the coefficients' share alone, S.ctxw[slot] the contexts behind those, and no cost is formed.  One plane per call, U then V. */
XA_DEV void nxn4_chroma_spec(const x265amd_intra_nxn_job& P, Nxn4Lds& S, const uint8_t* modes, int slot0, int slotEnd, int pl, const EnTabs& tabs, int lane, int grp, int l)
{
    const int slotRaw = slot0 + grp, active = slotRaw < slotEnd, m = active ? slotRaw : slotEnd - 1;
    const uint32_t mode = modes[m];
    const int scanType = mode >= 22 && mode <= 30 ? 1 : (mode >= 6 && mode <= 14 ? 2 : 0);
    const int y = l >> 2, x = l & 3;
    uint8_t* cw = S.ctxw[m];
    if (pl == 0)
    {
        if (active) for (int b = l; b < X265AMD_CTX_STRIDE; b += 16) cw[b] = P.ctx[b];
        if (active && l == 0) S.ccoef[m] = 0;
        xa_wave_sync();
    }
    const x265amd_intra_tu_job& C = P.ctmpl[pl];
    const Q4 qC = q4_make(C.tu.qp_scaled, C.tu.slice_type);
    const pixel* cr = S.cref[pl];
    const int part = l < 4 ? (int)cr[1 + l] + (int)cr[9 + l] : 0;
    int sdc = part;
    sdc += __builtin_amdgcn_update_dpp(0, sdc, 0xB1, 0xf, 0xf, true);
    sdc += __builtin_amdgcn_update_dpp(0, sdc, 0x4E, 0xf, 0xf, true);
    const int dc = (__shfl(sdc, lane & 48, 64) + 4) >> 3;
    const int f = S.cfenc[pl][l];
    const int p = nxn4_pred_sample(cr, S.csw[pl], S.tb, (int)mode, dc, y, x, false);
    const Chain4 ch = grp16_chain4(f, p, 0, qC, C.tu.sign_hide, scanType, S.tb, nxn4_energy(f, lane), lane);
    const int lvScan = __shfl(ch.lv, (lane & 48) + S.tb.scan[scanType][l], 64);
    if (active)
    {
        const uint32_t cf = ch.numSig ? grp16_coeff_bits4(cw, cw, lvScan, 0, scanType, C.tu.sign_hide, S.step, S.tb, lane) : 0u;
        S.crec[m][pl][l] = (pixel)ch.rec; S.clev[m][pl][l] = (int16_t)ch.lv;
        if (l == 0)
        {
            x265amd_tu_result r;
            r.num_sig = ch.numSig; r.zero_energy = ch.zeroEnergy; r.nz_energy = ch.nzEnergy; r.reserved = 0; r.zero_dist = ch.zeroDist; r.nz_dist = ch.nzDist;
            S.cres[m][pl] = r;
            S.ccoef[m] += cf;
        }
    }
    xa_wave_sync();
    if (pl == 1 && active && l == 0)
    {
        const x265amd_tu_result rU = S.cres[m][0], rV = S.cres[m][1];
        uint64_t frac = cb_bin_t(tabs, cw + CTX_QT_CBF + 2, rU.num_sig != 0 ? 1u : 0u);          /* the two coded block flags share a context */
        frac += cb_bin_t(tabs, cw + CTX_QT_CBF + 2, rV.num_sig != 0 ? 1u : 0u);
        S.cfrac[m] = frac + S.ccoef[m];
    }
}

/* a decided luma unit's share of the CU's bits on the running contexts: its coded block flag (C_QT_CBF + 0: one level down) and its coefficients -- one wavefront */
XA_DEV void nxn4_count_unit(const x265amd_intra_nxn_job& P, Nxn4Lds& S, int k, const EnTabs& tabs, int lane)
{
    const uint32_t cbf = S.ures[k].num_sig != 0;
    uint64_t add = 0;
    if (lane == 0) add = cb_bin_t(tabs, S.runCtx + CTX_QT_CBF, cbf);
    xa_wave_sync();
    if (cbf) add += wave_coeff_bits_4x4(S.runCtx, S.runCtx, S.lev + 16 * k, 0, 1, (int)S.winMode[k], P.tmpl[0].tu.sign_hide, S.step, lane);
    xa_wave_sync();
    if (lane == 0) S.runFrac += add;
    xa_wave_sync();
}

/* CUData::getAllowedChromaDir (cudata.cpp:889-907) -> the mode number as the CU stores it (36: derived) for place `idx` of the list */
XA_DEV uint32_t nxn4_chroma_stored(uint32_t lumaDir, uint32_t idx)
{
    uint32_t list[5] = { 0, 26, 10, 1, 36 };
    for (int i = 0; i < 4; i++) if (lumaDir == list[i]) { list[i] = 34; break; }
    return list[idx < 5 ? idx : 4];
}

/* The decision of a chained CU (role 1), with this workgroup's NxN evaluation in S and the other workgroup's 2Nx2N evaluation in its peer record: both CUs' bits as
 * Search::checkIntra counts them at its end (search.cpp:1254-1275), the costs (RDCost::calcRdCost / calcPsyRdCost, rdcost.h:89-123), the comparison of checkBestMode
 * in the order 2Nx2N, NxN (analysis.cpp:3670-3692: the later one only wins when strictly cheaper); then the winner's samples to the picture and the parent's tile, the
 * CU's result to the host, contexts / fraction / modes to the chain. */
XA_DEV void nxn4_decide(const x265amd_intra_nxn_job& P, Nxn4Lds& S, int tid, int nthr)
{
    const int lane = tid & 63, wv = tid >> 6;
    x265amd_intra_peer* peer = reinterpret_cast<x265amd_intra_peer*>(P.peer);
    x265amd_intra_chain* ch = reinterpret_cast<x265amd_intra_chain*>(P.chain);
    x265amd_intra_cu8_result* out = reinterpret_cast<x265amd_intra_cu8_result*>(P.cu_out);
    __syncthreads();
    XA_CHAIN_START();
    XA_LINK_T(5);
    if (tid == 0)
    {
        bool ok = xa_chain_wait(&peer->ready, P.chain_token + 1);
        if (ok && __hip_atomic_load(&peer->ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ~0ull) ok = false;      /* the other side gave up */
        S.peerOk = ok ? 1 : 0;
    }
    XA_LINK_T(6);
    __syncthreads();
    XA_CHAIN(2);
    if (!S.peerOk)
    {
        if (tid == 0) { out->status = 2; xa_chain_publish(&ch->seq, P.chain_token + 1); }
        return;
    }
    for (int i = tid; i < (int)(sizeof(x265amd_intra_nxn_out) / 8); i += nthr) reinterpret_cast<uint64_t*>(&S.peerOut)[i] = reinterpret_cast<const uint64_t*>(&peer->out)[i];
    /* both CUs' bits are counted already: the other CU's by its workgroup (the peer record), this one's by the last wavefront beside the evaluation -- what is left is
     * the chroma mode's share, which the chroma decision counted on contexts that are chroma's alone */
    const x265amd_intra_nxn_out& Q = S.peerOut;
    const uint32_t chromaN = nxn4_chroma_stored(S.winMode[0], S.cw), chroma2 = nxn4_chroma_stored(Q.mode[0], Q.chroma_best);
    for (int i = tid; i < X265AMD_CTX_STRIDE; i += nthr)
    {
        const uint8_t c = S.ctxw[S.cw][i];
        S.fctx[0][i] = c != P.ctx[i] ? c : S.runCtx[i];
        S.fctx[1][i] = peer->fctx[i];
    }
    if (tid == 0)
    {
        S.ffrac[0] = S.runFrac + (S.cfrac[S.cw] - P.scan_frac);
        S.fmv[0] = (uint64_t)P.scan_frac + S.runMv + S.enBits[P.ctx[14] ^ (chromaN == 36 ? 0u : 1u)] + (chromaN != 36 ? (2ull << 15) : 0ull);
        S.ffrac[1] = peer->ffrac; S.fmv[1] = peer->fmv;
    }
    __syncthreads();
    if (P.reserved[0])
    {
        /* cu_qp_delta (Entropy::codeDeltaQP, entropy.cpp:1737-1756) behind the coded block flags of an evaluation that has coefficients: its bins' contexts (16, 17) are nobody
         * else's, so adding them here is adding them where the reference codes them */
        if (tid == 0)
        {
            const int dqp = (int)(int8_t)P.reserved[1];
            const uint32_t a = (uint32_t)(dqp < 0 ? -dqp : dqp);
            bool any[2];
            any[0] = S.ures[0].num_sig || S.ures[1].num_sig || S.ures[2].num_sig || S.ures[3].num_sig || S.cres[S.cw][0].num_sig || S.cres[S.cw][1].num_sig;
            any[1] = Q.res[0].num_sig || Q.cres[0].num_sig || Q.cres[1].num_sig;
            for (int c = 0; c < 2; c++)
            {
                if (!any[c]) continue;
                uint8_t* cx = S.fctx[c];
                uint64_t f = S.ffrac[c];
                const uint32_t first = a ? 1u : 0u;
                f += cb_bin(&cx[16], first);
                if (a)
                {
                    /* writeUnaryMaxSymbol(min(a, 5), ctx 16, offset 1, max 5) */
                    uint32_t sym = a < 5 ? a : 5;
                    const bool codeLast = 5 > sym;
                    while (--sym) f += cb_bin(&cx[17], 1u);
                    if (codeLast) f += cb_bin(&cx[17], 0u);
                    if (a >= 5)
                    {
                        /* writeEpExGolomb(a - 5, 0): bypass bins */
                        uint32_t symbol = a - 5, count = 0, n = 0;
                        while (symbol >= (1u << count)) { n++; symbol -= 1u << count; count++; }
                        n += 1 + count;
                        f += (uint64_t)32768 * n;
                    }
                    f += 32768;         /* the sign */
                }
                S.ffrac[c] = f;
            }
        }
        __syncthreads();
    }
    XA_CHAIN(3);
    /* the two costs */
    const uint32_t lumaN = (uint32_t)(S.ures[0].nz_dist + S.ures[1].nz_dist + S.ures[2].nz_dist + S.ures[3].nz_dist);
    const uint32_t chromaDN = (uint32_t)(S.cres[S.cw][0].nz_dist + S.cres[S.cw][1].nz_dist), chromaD2 = (uint32_t)(Q.cres[0].nz_dist + Q.cres[1].nz_dist);
    const uint32_t luma2 = (uint32_t)Q.res[0].nz_dist;
    const uint32_t psyN = P.psy_scale ? S.psyNxn : 0u, psy2 = P.psy_scale ? Q.res[0].nz_energy : 0u;
    const uint32_t bitsN = (uint32_t)(S.ffrac[0] >> 15), bits2 = (uint32_t)(S.ffrac[1] >> 15);
    const uint64_t distN = (uint64_t)lumaN + chromaDN, dist2 = (uint64_t)luma2 + chromaD2;        /* sse_t sums: an 8x8 CU's stay far below 32 bits at either depth */
    const uint64_t costN = P.psy_scale ? distN + ((P.psy_scale * (uint64_t)psyN) >> 24) + (((uint64_t)bitsN * P.lambda2) >> 8) : distN + (((uint64_t)bitsN * P.lambda2 + 128) >> 8);
    const uint64_t cost2 = P.psy_scale ? dist2 + ((P.psy_scale * (uint64_t)psy2) >> 24) + (((uint64_t)bits2 * P.lambda2) >> 8) : dist2 + (((uint64_t)bits2 * P.lambda2 + 128) >> 8);
    const bool nxnWins = costN < cost2;
    const int win = nxnWins ? 0 : 1;
    /* the winner's samples: the picture and the parent's tile */
    {
        const x265amd_intra_tu_job& T0 = P.tmpl[0];
        pixel* pic = reinterpret_cast<pixel*>(T0.nb);
        pixel* dstY = reinterpret_cast<pixel*>(P.win_dst[0]);
        const pixel* peerY = reinterpret_cast<const pixel*>(P.peer_recon[0]);
        if (tid < 64)
        {
            const int y = tid >> 3, x = tid & 7;
            const pixel v = nxnWins ? S.frame[(y + 1) * 17 + x + 1] : peerY[y * 64 + x];
            pic[(long)y * T0.nb_stride + x] = v;
            dstY[y * 64 + x] = v;
        }
        else if (tid < 96)
        {
            const int pl = (tid - 64) >> 4, i = tid & 15, y = i >> 2, x = i & 3;
            const x265amd_intra_tu_job& C = P.ctmpl[pl];
            const pixel v = nxnWins ? S.crec[S.cw][pl][i] : reinterpret_cast<const pixel*>(P.peer_recon[1 + pl])[y * 32 + x];
            reinterpret_cast<pixel*>(C.nb)[(long)y * C.nb_stride + x] = v;
            reinterpret_cast<pixel*>(P.win_dst[1 + pl])[y * 32 + x] = v;
        }
        else if (tid >= 192 && tid < 192 + X265AMD_CTX_STRIDE) ch->ctx[tid - 192] = S.fctx[win][tid - 192];
    }
    /* what the next CU of the chain takes is out first, and the chain goes on; the host's record (pinned host memory: its stores cross the bus, and the release in front
     * of the chain's word would wait for every one of them) is written behind that -- the host reads it when the command is over */
    if (tid == 0)
    {
        for (int k = 0; k < 4; k++) ch->mode[P.chain_index & 3][k] = nxnWins ? S.winMode[k] : Q.mode[0];
        ch->frac = S.ffrac[win];
    }
    __syncthreads();
    XA_CHAIN(4);
    if (tid == 0) xa_chain_publish(&ch->seq, P.chain_token + 1);
    XA_CHAIN(5);
    XA_LINK_T(4);
    if (tid >= 96 && tid < 96 + 96)
    {
        const int i = tid - 96;
        int16_t v;
        if (i < 64) v = nxnWins ? S.lev[i] : Q.levels[0][i];
        else v = nxnWins ? S.clev[S.cw][(i - 64) >> 4][i & 15] : Q.clevels[(i - 64) >> 4][i & 15];
        out->levels[i] = v;
    }
    else if (tid >= 192 && tid < 192 + X265AMD_CTX_STRIDE) out->ctx[tid - 192] = S.fctx[win][tid - 192];
    if (tid == 0)
    {
        out->rd_cost = nxnWins ? costN : cost2; out->other_cost = nxnWins ? cost2 : costN; out->frac_bits = S.ffrac[win];
        const uint32_t tb = nxnWins ? bitsN : bits2, mvb = (uint32_t)(S.fmv[win] >> 15);
        out->total_bits = tb; out->mv_bits = mvb; out->coeff_bits = tb - mvb;
        out->psy_energy = nxnWins ? psyN : psy2; out->res_energy = nxnWins ? S.resNxn : (uint32_t)Q.res[0].zero_dist;
        out->luma_dist = nxnWins ? lumaN : luma2; out->chroma_dist = nxnWins ? chromaDN : chromaD2;
        out->part_size = nxnWins ? 3 : 0; out->chroma_dir = (uint8_t)(nxnWins ? chromaN : chroma2);
        out->cbf_u = nxnWins ? (S.cres[S.cw][0].num_sig != 0) : (Q.cres[0].num_sig != 0); out->cbf_v = nxnWins ? (S.cres[S.cw][1].num_sig != 0) : (Q.cres[1].num_sig != 0);
        for (int k = 0; k < 4; k++)
        {
            const uint8_t m = nxnWins ? S.winMode[k] : Q.mode[0];
            out->luma_dir[k] = m;
            out->cbf_y[k] = nxnWins ? (S.ures[k].num_sig != 0) : (Q.res[0].num_sig != 0);
        }
        out->status = 1;
    }
}

/* rqBase: NXN4_RQ_BYTES of LDS behind S when the command quantises with RDOQ */
XA_DEV void block_intra_nxn4(const x265amd_intra_nxn_job& P, x265amd_intra_nxn_out* po, Nxn4Lds& S, IntraPuShared& L, int tid, int nthr, char* rqBase = nullptr)
{
    const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6, l = lane & 15, grp = lane >> 4;
    const x265amd_intra_tu_job& T0 = P.tmpl[0];
    const long ps = T0.nb_stride;
    pixel* pic = reinterpret_cast<pixel*>(T0.nb);                   /* the CU's first sample in the reconstructed plane */
    const bool chained = P.chain != 0 && P.chain_role == 1;          /* the picture, the tiles and the result wait for the decision between this and the other evaluation */
    const bool rdoq = P.rdoq_level != 0;
    XA_NXN_START(0);
    if (rdoq)
    {
        for (int i = tid; i < 2 * 184; i += nthr) (&S.est[0][0])[i] = 0;
        if (tid == 0) S.rq = reinterpret_cast<Rq4Area*>(rqBase);
        __syncthreads();
        if (wv == 0) wave_est_bit(P.ctx, S.est[0], 2, 1, lane);
        else if (wv == 1) wave_est_bit(P.ctx, S.est[1], 2, 0, lane);
    }
    /* ---- tables, the source block, the neighbourhood ---- */
    nxn4_fill_tabs(S.tb, tid);
    if (tid < 128) S.enBits[tid] = en_bits[tid];
    if (tid < 64) S.enLps[tid] = en_lpsNext[tid];
    for (int i = tid; i < 256; i += nthr) S.step[i] = en_step.v[i];
    if (tid >= 128 && tid < 192)
    {
        const int i = tid - 128;
        S.fenc[i] = reinterpret_cast<const pixel*>(T0.tu.fenc)[(i >> 3) * T0.tu.fenc_stride + (i & 7)];
    }
    if (tid >= 192 && tid < 192 + 33)
    {
        /* the ring: above-left, twelve above, twelve left -- a segment is read when a unit that looks at it says it is there */
        const int i = tid - 192;
        const uint32_t a0 = (uint32_t)P.tmpl[0].avail, a1 = (uint32_t)P.tmpl[1].avail, a2 = (uint32_t)P.tmpl[2].avail;
        bool have; int x, y;
        if (i == 0) { x = -1; y = -1; have = (a0 >> 2) & 1; }
        else if (i <= 16) { x = i - 1; y = -1; have = x < 4 ? ((a0 >> 3) & 1) : (x < 8 ? (((a0 >> 4) | (a1 >> 3)) & 1) : (x < 12 ? ((a1 >> 4) & 1) : false)); }
        else { x = -1; y = i - 17; have = y < 4 ? ((a0 >> 1) & 1) : (y < 8 ? ((a0 | (a2 >> 1)) & 1) : (y < 12 ? (a2 & 1) : false)); }
        S.frame[(y + 1) * 17 + x + 1] = have ? pic[(long)y * ps + x] : (pixel)0;
    }
    if (P.do_chroma && wv >= nwv - 2)
    {
        /* the two chroma blocks' neighbours and source samples (initAdiPatternChroma: no smoothing): a wavefront per plane, so that their reads of the picture are one
         * round trip and not two (one wavefront alone takes them one after the other) */
        for (int pl = nwv >= 2 ? wv - (nwv - 2) : 0; pl < (nwv >= 2 ? wv - (nwv - 2) + 1 : 2); pl++)
        {
            const x265amd_intra_tu_job& C = P.ctmpl[pl];
            (void)nxn4_neighbours(reinterpret_cast<const pixel*>(C.nb), (int)C.nb_stride, (uint32_t)C.avail, S.cref[pl], S.csw[pl], lane);
            if (lane < 16) S.cfenc[pl][lane] = reinterpret_cast<const pixel*>(C.tu.fenc)[(lane >> 2) * C.tu.fenc_stride + (lane & 3)];
        }
    }
    if (tid == 0) XA_BYTES((64 + 33 + 2 * (16 + 17)) * sizeof(pixel) + (3 * 64 + 2 * 2 * 16) * sizeof(pixel) + (64 + 32) * 2 + sizeof(x265amd_intra_nxn_job) + sizeof(x265amd_intra_nxn_out));
    const EnTabs tabs{ S.enBits, S.enLps };
    const uint32_t adi = P.ctx[13];                                                         /* C_ADI: prev_intra_luma_pred_flag */
    const uint32_t rbits = (uint32_t)(((unsigned long long)P.scan_frac + en_bits[adi ^ 0]) >> 15) + 5;
    const uint32_t mpmBase = (uint32_t)(((unsigned long long)P.scan_frac + en_bits[adi ^ 1]) >> 15);
    const int maxCand = P.max_cand > 16 ? 16 : P.max_cand;
    const Q4 qY = q4_make(T0.tu.qp_scaled, T0.tu.slice_type);
    const int signHide = T0.tu.sign_hide;
    __syncthreads();
    XA_NXN(0);
    pixel* ref = S.ref[wv]; pixel* sw = S.sw[wv];
    /* a chained CU keeps its last wavefront out of the evaluation: it counts the decided units' bits on the running contexts while the others go on (the coefficient
     * contexts of the luma units move from unit to unit, so the candidates' estimates -- all from the start contexts -- do not serve) */
    /* Eight wavefronts: five evaluate the luma units (twenty groups: the 35 modes of the scan in two rounds, up to twenty candidates side by side), two run the
     * chroma decision beside luma units 1 and 2 (its mode list needs only the first unit's direction), the last one counts bits when the CU is chained.  Fewer
     * wavefronts: all of them on the luma units, the chroma decision behind them. */
    const bool overlap = nwv >= 8 && P.do_chroma;
    const int lw = overlap ? 5 : (chained ? nwv - 1 : nwv);
    const bool worker = wv < lw;
    const bool chromaWave = overlap && (wv == 5 || wv == 6);
    if (chained)
    {
        for (int i = tid; i < X265AMD_CTX_STRIDE; i += nthr) S.runCtx[i] = P.ctx[i];
        if (tid == 0) { S.runFrac = P.scan_frac; S.runMv = 0; }
        __syncthreads();
    }
    for (int k = 0; k < 4; k++)
    {
        const int ux = 4 * (k & 1), uy = 4 * (k >> 1);
        const pixel* org = S.frame + (uy + 1) * 17 + ux + 1;
        /* getIntraDirLumaPredictor: the left / above units' modes */
        const uint32_t left = (k & 1) ? S.winMode[k - 1] : P.left_mode[k >> 1], above = (k & 2) ? S.winMode[k - 2] : P.above_mode[k & 1];
        uint32_t p0, p1, p2;
        if (left == above)
        {
            if (left >= 2) { p0 = left; p1 = ((left - 2 + 31) & 31) + 2; p2 = ((left - 2 + 1) & 31) + 2; }
            else { p0 = 0; p1 = 1; p2 = 26; }
        }
        else { p0 = left; p1 = above; p2 = (left && above) ? 0 : ((left + above) < 2 ? 26 : 1); }
        if (tid == 0) { S.preds[k][0] = (uint8_t)p0; S.preds[k][1] = (uint8_t)p1; S.preds[k][2] = (uint8_t)p2; }
        /* every wavefront gathers the unit's neighbours for itself */
        const int dc = nxn4_neighbours(org, 17, (uint32_t)P.tmpl[k].avail, ref, sw, lane);
        xa_wave_sync();
        XA_NXN(1);
        const int y = l >> 2, x = l & 3;
        const int f = S.fenc[(uy + y) * 8 + ux + x];
        const int srcEnergy = nxn4_energy(f, lane);
        const RdoqParams rpL = { S.est[0], P.rdoq_lambda2[0], P.rdoq_lambda[0], P.psy_rdoq_scale, P.rdoq_level, P.rdoq_tu_depth };
        const int fD = rdoq && P.psy_rdoq_scale ? grp16_forward4(f, S.tb.T[0], lane) : 0;
        /* the scan: a group per mode, SATD of the residual (cu[4x4].sa8d = satd_4x4, pixel.cpp:1171) */
        for (int m0 = 0; m0 < 35 && worker; m0 += 4 * lw)
        {
            const int mode = m0 + wv * 4 + grp, m = mode < 35 ? mode : 34;
            int v = f - nxn4_pred_sample(ref, sw, S.tb, m, dc, y, x, true);
            v = xa_lane_had4x4(v, lane);
            v = xa_row16_sum(abs(v));
            if (l == 0 && mode < 35) L.sa8d[mode] = v >> 1;
        }
        __syncthreads();
        XA_NXN(2);
        if (tid < 64) wave0_candidate_list(L, p0, p1, p2, rbits, mpmBase, P.lambda, maxCand, tid);
        __syncthreads();
        XA_NXN(3);
        const int n = L.num;
        /* (a chained CU's last wavefront: the unit before this one joins the CU's bit count while the candidates run -- the longest stretch between two barriers) */
        if (chained && wv == nwv - 1 && k > 0) nxn4_count_unit(P, S, k - 1, tabs, lane);
        if (chromaWave && (k == 1 || k == 2))
        {
            const uint32_t lumaDir0 = S.winMode[0];
            uint32_t clist[5] = { 0, 26, 10, 1, 36 };               /* CUData::getAllowedChromaDir (cudata.cpp:889-907) */
            for (int i = 0; i < 4; i++) if (lumaDir0 == clist[i]) { clist[i] = 34; break; }
            nxn4_chroma_plane(P, S, clist, lumaDir0, k - 1, tabs, lane, wv - 5, grp, l, rdoq ? S.rq + wv * 4 + grp : nullptr);
        }
        /* the candidates' chains: candidate c = group * waves + wavefront (the first eight one per wavefront) */
        Chain4 mine = {};
        int myCand = -1;
        for (int c0 = 0; c0 < n && worker; c0 += 4 * lw)
        {
            const int c = c0 + grp * lw + wv;
            const int cc = c < n ? c : n - 1;
            const uint32_t mode = L.modes[cc];
            const int p = nxn4_pred_sample(ref, sw, S.tb, (int)mode, dc, y, x, true);
            const int scanType = mode >= 22 && mode <= 30 ? 1 : (mode >= 6 && mode <= 14 ? 2 : 0);
            Chain4 ch;
            if (rdoq)
            {
                const Rq4 rq = { S.rq + wv * 4 + grp, &rpL, 0, (int)mode, T0.tu.qp_scaled, fD };
                ch = grp16_chain4<true>(f, p, 1, qY, signHide, scanType, S.tb, srcEnergy, lane, &rq);
            }
            else ch = grp16_chain4(f, p, 1, qY, signHide, scanType, S.tb, srcEnergy, lane);
            const int lvScan = __shfl(ch.lv, (lane & 48) + S.tb.scan[scanType][l], 64);
            const uint32_t coeffFrac = ch.numSig ? grp16_coeff_bits4(P.ctx, nullptr, lvScan, 1, scanType, signHide, S.step, S.tb, lane) : 0u;
            if (c < n)
            {
                mine = ch; myCand = c;
                if (l == 0)
                {
                    /* the candidate's bits and cost (codeIntraLumaQT, search.cpp:357-400) */
                    const uint8_t* cw = P.ctx;
                    const int pidx = mode == p0 ? 0 : (mode == p1 ? 1 : (mode == p2 ? 2 : -1));
                    unsigned long long frac = P.frac_start[k];
                    frac += S.enBits[cw[13] ^ (pidx != -1 ? 1u : 0u)];
                    frac += (unsigned long long)(pidx != -1 ? 1 + (pidx != 0) : 5) << 15;
                    frac += S.enBits[cw[CTX_QT_CBF] ^ (ch.numSig != 0 ? 1u : 0u)];
                    frac += coeffFrac;
                    const unsigned long long bits = (uint32_t)(frac >> 15);
                    const unsigned long long dist = ch.nzDist;
                    S.cost[c] = P.psy_scale ? dist + ((P.psy_scale * (unsigned long long)ch.nzEnergy) >> 24) + ((bits * P.lambda2) >> 8) : dist + ((bits * P.lambda2 + 128) >> 8);
                }
            }
        }
        XA_NXN(4);
        __syncthreads();
        XA_NXN(5);
        /* the choice: the first cheapest; its group puts the unit in place */
        int w = 0;
        {
            unsigned long long best = ~0ull;
            for (int i = 0; i < n; i++) { const unsigned long long cst = S.cost[i]; if (cst < best) { best = cst; w = i; } }
        }
        if (myCand == w)
        {
            S.frame[(uy + 1 + y) * 17 + ux + 1 + x] = (pixel)mine.rec;
            S.pred[(uy + y) * 8 + ux + x] = (pixel)mine.pred;
            S.lev[k * 16 + l] = (int16_t)mine.lv;
            if (l == 0)
            {
                const uint8_t mode = L.modes[w];
                S.winMode[k] = mode;
                x265amd_tu_result r;
                r.num_sig = mine.numSig; r.zero_energy = mine.zeroEnergy; r.nz_energy = mine.nzEnergy; r.reserved = 0; r.zero_dist = mine.zeroDist; r.nz_dist = mine.nzDist;
                S.ures[k] = r;
                if (!chained) { po->mode[k] = mode; po->num_cand[k] = (uint8_t)n; po->res[k] = r; }
            }
        }
        __syncthreads();
        XA_NXN(6);
    }
    XA_NXN(7);
    /* ---- the CU's luma: picture, tiles, levels, the two measurements ---- */
    if (tid < 64)
    {
        const int y = tid >> 3, x = tid & 7, k = (y >> 2) * 2 + (x >> 2), yy = y & 3, xx = x & 3;
        if (!chained)
        {
            const pixel v = S.frame[(y + 1) * 17 + x + 1];
            pic[(long)y * ps + x] = v;
            reinterpret_cast<pixel*>(P.layer_dst[k])[yy * 64 + xx] = v;
            if (P.recon_dst[k]) reinterpret_cast<pixel*>(P.recon_dst[k])[yy * 64 + xx] = v;
            reinterpret_cast<pixel*>(P.pred_dst[k])[yy * 64 + xx] = S.pred[y * 8 + x];
            int16_t* lvOut = P.levels_dst ? reinterpret_cast<int16_t*>(P.levels_dst) : &po->levels[0][0];
            lvOut[tid] = S.lev[tid];
        }
        const int psy = wave_psy_cost(S.fenc, 8, S.frame + 18, 17, 1, lane);
        const uint64_t sse = wave_sse_pp(S.fenc, 8, S.pred, 8, 8, lane);
        if (lane == 0)
        {
            S.psyNxn = (uint32_t)psy; S.resNxn = (uint32_t)sse;
            if (!chained) { po->psy_energy = (uint32_t)psy; po->res_energy = (uint32_t)sse; }
        }
    }
    XA_NXN(8);
    if (!P.do_chroma) return;
    /* ---- estIntraPredChromaQT for the one 4x4 block per plane (search.cpp:1754-1889): a group per mode runs U, then V on the contexts U has moved ---- */
    {
        const uint32_t lumaDir = S.winMode[0];
        uint32_t list[5] = { 0, 26, 10, 1, 36 };                /* CUData::getAllowedChromaDir (cudata.cpp:889-907) */
        for (int i = 0; i < 4; i++) if (lumaDir == list[i]) { list[i] = 34; break; }
        if (chained && wv == nwv - 1)
        {
            /* the last luma unit, then the bins in front of the transform tree: partition size, the four prev_intra_luma_pred_flags and the mode bits (codePredInfo) */
            nxn4_count_unit(P, S, 3, tabs, lane);
            if (lane == 0)
            {
                uint8_t* cw = S.runCtx;
                uint64_t pi = cb_bin_t(tabs, cw + 8, 0u);                           /* C_PART_SIZE: NxN */
                for (int j = 0; j < 4; j++)
                {
                    const uint32_t d = S.winMode[j];
                    const int pidx = d == S.preds[j][0] ? 0 : (d == S.preds[j][1] ? 1 : (d == S.preds[j][2] ? 2 : -1));
                    pi += cb_bin_t(tabs, cw + 13, pidx != -1 ? 1u : 0u);             /* C_ADI: the four flags one after the other */
                    pi += (uint64_t)(pidx != -1 ? 1 + (pidx != 0) : 5) << 15;
                }
                /* S.runMv: what the luma prediction info adds to the start fraction (the chroma mode's share joins when it is known) */
                S.runMv = pi; S.runFrac += pi;
            }
        }
        if (!overlap) nxn4_chroma_modes(P, S, list, lumaDir, tabs, lane, wv, grp, l);
        __syncthreads();
        int w = 0;
        {
            unsigned long long best = ~0ull;
            for (int i = 0; i < 5; i++) if (S.ccost[i] < best) { best = S.ccost[i]; w = i; }
        }
        if (tid == 0)
        {
            S.cw = (uint32_t)w;
            if (!chained) { po->chroma_best = (uint32_t)w; po->chroma_reserved = 0; po->cres[0] = S.cres[w][0]; po->cres[1] = S.cres[w][1]; }
        }
        if (tid < 32 && !chained)
        {
            const int pl = tid >> 4, i = tid & 15, y = i >> 2, x = i & 3;
            const x265amd_intra_tu_job& C = P.ctmpl[pl];
            int16_t* clOut = P.clevels_dst ? reinterpret_cast<int16_t*>(P.clevels_dst) : &po->clevels[0][0];
            reinterpret_cast<pixel*>(P.crecon_dst[pl])[y * 32 + x] = S.crec[w][pl][i];
            reinterpret_cast<pixel*>(C.nb)[(long)y * C.nb_stride + x] = S.crec[4][pl][i];       /* the picture keeps the LAST tried mode's samples, as after the reference's loop */
            clOut[pl * 16 + i] = S.clev[w][pl][i];
        }
    }
    XA_NXN(9);
    if (chained) nxn4_decide(P, S, tid, nthr);
}

#endif
