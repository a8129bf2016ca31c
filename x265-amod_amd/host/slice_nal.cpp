/* Slice NAL units (include/x265amd.h: x265amd_write_slice_nal): host C++, part of SURVEY section 8f rank 4 (formats).
 *
 * Restatement of Entropy::codeSliceHeader / codeSliceHeaderWPPEntryPoints / codeShortTermRefPicSet (reference:
 * source/encoder/entropy.cpp:593-766), Bitstream::writeByteAlignment (source/common/bitstream.cpp), and the byte-stream packing of
 * NALList::serialize / serializeSubstreams (source/encoder/nal.cpp:60-232): start code, 16-bit NAL header, emulation prevention --
 * the header and the sub-streams are escaped separately, exactly as the reference does (the escape state does not carry from the
 * header into the slice data). */
#include "x265amd.h"
#include <string.h>
#include <vector>

namespace {

struct Bits
{
    std::vector<uint8_t> out; uint32_t partial = 0; int n = 0;
    void put(uint32_t val, int bits) { for (int i = bits - 1; i >= 0; i--) { partial = (partial << 1) | ((val >> i) & 1); if (++n == 8) { out.push_back((uint8_t)partial); partial = 0; n = 0; } } }
    void flag(bool f) { put(f ? 1 : 0, 1); }
    void ue(uint32_t v) { uint32_t len = 1, t = ++v; while (t > 1) { t >>= 1; len += 2; } put(0, (int)(len >> 1)); put(v, (int)((len + 1) >> 1)); }
    void se(int v) { ue(v <= 0 ? (uint32_t)(-v) << 1 : ((uint32_t)v << 1) - 1); }
    void align() { put(1, 1); while (n) put(0, 1); }
};

}

extern "C" size_t x265amd_write_slice_nal(const x265amd_slice_header* h, const uint8_t* substreams, const uint32_t* sizes, int numSubstreams, uint8_t* out, size_t cap)
{
    if (!h || numSubstreams < 0 || (numSubstreams && (!substreams || !sizes))) return 0;
    /* sub-streams first: their escaped sizes go into the header (serializeSubstreams) */
    std::vector<uint8_t> data;
    std::vector<uint32_t> escaped((size_t)(numSubstreams > 0 ? numSubstreams : 1), 0);
    uint32_t maxSize = 0;
    const uint8_t* in = substreams;
    for (int s = 0; s < numSubstreams; s++)
    {
        const size_t before = data.size();
        for (uint32_t i = 0; i < sizes[s]; i++)
        {
            const size_t b = data.size();
            if (b >= 2 && !data[b - 2] && !data[b - 1] && in[i] <= 3) data.push_back(3);
            data.push_back(in[i]);
        }
        in += sizes[s];
        if (s < numSubstreams - 1)
        {
            escaped[s] = (uint32_t)(data.size() - before);
            if (escaped[s] > maxSize) maxSize = escaped[s];
        }
    }
    const bool rap = h->nal_unit_type >= 16 && h->nal_unit_type <= 23, idr = h->nal_unit_type == 19 || h->nal_unit_type == 20;
    const int sliceType = h->slice_type;         /* 0 B, 1 P, 2 I as in the bitstream */
    Bits b;
    b.flag(true);                                /* first_slice_segment_in_pic_flag (one slice per picture) */
    if (rap) b.flag(false);                      /* no_output_of_prior_pics_flag */
    b.ue(0);                                     /* slice_pic_parameter_set_id */
    b.ue((uint32_t)sliceType);
    if (!idr)
    {
        const int lsbBits = h->log2_max_poc_lsb;
        b.put((uint32_t)((h->poc - h->last_idr_poc + (1 << lsbBits)) % (1 << lsbBits)), lsbBits);
        if (h->rps_idx < 0)
        {
            b.flag(false);                       /* short_term_ref_pic_set_sps_flag */
            if (h->num_rps_in_sps > 0) b.flag(false);            /* inter_ref_pic_set_prediction_flag (idx > 0) */
            b.ue((uint32_t)h->num_negative); b.ue((uint32_t)h->num_positive);
            int prev = 0;
            for (int j = 0; j < h->num_negative; j++) { b.ue((uint32_t)(prev - h->delta_poc[j] - 1)); prev = h->delta_poc[j]; b.flag(h->used[j] != 0); }
            prev = 0;
            for (int j = h->num_negative; j < h->num_negative + h->num_positive; j++) { b.ue((uint32_t)(h->delta_poc[j] - prev - 1)); prev = h->delta_poc[j]; b.flag(h->used[j] != 0); }
        }
        else
        {
            b.flag(true);
            int numBits = 0;
            while ((1 << numBits) < h->num_rps_in_sps) numBits++;
            if (numBits > 0) b.put((uint32_t)h->rps_idx, numBits);
        }
        if (h->temporal_mvp_enabled) b.flag(true);               /* slice_temporal_mvp_enable_flag */
    }
    if (h->use_sao) { b.flag(h->sao_luma != 0); b.flag(h->sao_chroma != 0); }
    else if (h->selective_sao) { b.flag(false); b.flag(false); }
    if (sliceType != 2)
    {
        const bool over = h->num_ref_idx[0] != h->num_ref_idx_default[0] || (sliceType == 0 && h->num_ref_idx[1] != h->num_ref_idx_default[1]);
        b.flag(over);
        if (over) { b.ue((uint32_t)(h->num_ref_idx[0] - 1)); if (sliceType == 0) b.ue((uint32_t)(h->num_ref_idx[1] - 1)); }
    }
    if (sliceType == 0) b.flag(false);           /* mvd_l1_zero_flag */
    if (h->temporal_mvp_enabled)
    {
        if (sliceType == 0) b.flag(h->col_from_l0 != 0);
        if (sliceType != 2 && ((h->col_from_l0 && h->num_ref_idx[0] > 1) || (!h->col_from_l0 && h->num_ref_idx[1] > 1))) b.ue((uint32_t)h->col_ref_idx);
    }
    if (sliceType != 2) b.ue((uint32_t)(5 - h->max_num_merge_cand));
    b.se(h->slice_qp - h->pps_init_qp);
    if (h->chroma_qp_offsets_present) { b.se(h->cb_qp_offset); b.se(h->cr_qp_offset); }
    {
        const bool saoOn = h->use_sao && (h->sao_luma || h->sao_chroma), dbfOn = !h->deblocking_disabled;
        if (saoOn || dbfOn) b.flag(h->slfase_flag != 0);
    }
    if (h->wpp)
    {
        const uint32_t n = numSubstreams > 0 ? (uint32_t)numSubstreams - 1 : 0;
        uint32_t offsetLen = 1;
        while (maxSize >= (1u << offsetLen)) offsetLen++;
        b.ue(n);
        if (n > 0) b.ue(offsetLen - 1);
        for (uint32_t i = 0; i < n; i++) b.put(escaped[i] - 1, (int)offsetLen);
    }
    b.align();
    /* NALList::serialize */
    std::vector<uint8_t> nal;
    if (h->first_in_access_unit) nal.push_back(0);
    nal.push_back(0); nal.push_back(0); nal.push_back(1);
    const size_t hdr0 = nal.size();
    nal.push_back((uint8_t)(h->nal_unit_type << 1));
    nal.push_back((uint8_t)(h->temporal_id_plus1 ? h->temporal_id_plus1 : 1));
    for (size_t i = 0; i < b.out.size(); i++)
    {
        const size_t bytes = nal.size();
        if (i > 2 && !nal[bytes - 2] && !nal[bytes - 3] && nal[bytes - 1] <= 3)
        {
            const uint8_t last = nal[bytes - 1];
            nal[bytes - 1] = 3;
            nal.push_back(last);
        }
        nal.push_back(b.out[i]);
    }
    (void)hdr0;
    nal.insert(nal.end(), data.begin(), data.end());
    if (!nal.back()) nal.push_back(3);
    if (out && cap >= nal.size()) memcpy(out, nal.data(), nal.size());
    return nal.size();
}
