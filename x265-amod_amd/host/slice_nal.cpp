/* Slice NAL units (include/x265amd.h: x265amd_write_slice_nal, x265amd_write_stream_headers): host C++, part of SURVEY section 8f rank 4 (formats).
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

/* NALList::serialize (nal.cpp:60-160): start code, NAL header, payload with emulation prevention (the look-back starts past the header) */
void serialize(std::vector<uint8_t>& nal, int nalUnitType, int temporalIdPlus1, bool longStartCode, const std::vector<uint8_t>& payload)
{
    if (longStartCode) nal.push_back(0);
    nal.push_back(0); nal.push_back(0); nal.push_back(1);
    nal.push_back((uint8_t)(nalUnitType << 1));
    nal.push_back((uint8_t)temporalIdPlus1);
    for (size_t i = 0; i < payload.size(); i++)
    {
        const size_t bytes = nal.size();
        if (i > 2 && !nal[bytes - 2] && !nal[bytes - 3] && nal[bytes - 1] <= 3)
        {
            const uint8_t last = nal[bytes - 1];
            nal[bytes - 1] = 3;
            nal.push_back(last);
        }
        nal.push_back(payload[i]);
    }
}

/* Entropy::codeProfileTier (entropy.cpp:380-429) */
void profileTier(Bits& b, const x265amd_stream_params* p)
{
    b.put(0, 2);
    b.flag(p->tier_flag != 0);
    b.put((uint32_t)p->profile_idc, 5);
    for (int j = 0; j < 32; j++) b.flag((p->profile_compatibility_flags >> j) & 1);
    b.flag(p->progressive_source != 0); b.flag(p->interlaced_source != 0); b.flag(p->non_packed_constraint != 0); b.flag(p->frame_only_constraint != 0);
    if (p->profile_idc == 4 || p->profile_idc == 5)            /* MAINREXT, HIGHTHROUGHPUTREXT */
    {
        const int depth = p->bit_depth_constraint, csp = p->chroma_format_constraint;     /* csp: 0 400, 1 420, 2 422, 3 444 */
        b.flag(depth <= 12); b.flag(depth <= 10); b.flag(depth <= 8 && csp != 2);
        b.flag(csp == 2 || csp == 1 || csp == 0); b.flag(csp == 1 || csp == 0); b.flag(csp == 0);
        b.flag(p->intra_constraint != 0); b.flag(p->one_picture_only_constraint != 0); b.flag(p->lower_bit_rate_constraint != 0);
        b.put(0, 16); b.put(0, 16); b.put(0, 3);
    }
    else { b.put(0, 16); b.put(0, 16); b.put(0, 12); }
    b.put((uint32_t)p->level_idc, 8);
    if (p->max_temporal_sub_layers > 1)
    {
        for (int i = 0; i < p->max_temporal_sub_layers - 1; i++) { b.flag(false); b.flag(false); }
        for (int i = p->max_temporal_sub_layers - 1; i < 8; i++) b.put(0, 2);
    }
}

}

/* The user-data SEI NAL unit that names the encoder and its options (Encoder::getStreamHeaders, encoder.cpp:3260-3280; SEIuserDataUnregistered, sei.h:89-117; SEI::writeSEImessages,
 * sei.cpp:35-75): prefix SEI (NAL type 39), payload type 5, the reference's UUID, the text, rbsp trailing bits.  Returns the unit's size behind a 4-byte start code, 0 when
 * it does not fit. */
extern "C" size_t x265amd_write_info_sei(const char* text, uint8_t* out, size_t cap)
{
    if (!text || !out) return 0;
    static const uint8_t uuid[16] = { 0x2C, 0xA2, 0xDE, 0x09, 0xB5, 0x17, 0x47, 0xDB, 0xBB, 0x55, 0xA4, 0xFE, 0x7F, 0xC2, 0xFC, 0x4E };
    const size_t len = strlen(text);
    Bits b;
    b.put(5, 8);                                    /* payload type: user_data_unregistered */
    size_t size = 16 + len;
    for (; size >= 0xff; size -= 0xff) b.put(0xff, 8);
    b.put((uint32_t)size, 8);
    for (int i = 0; i < 16; i++) b.put(uuid[i], 8);
    for (size_t i = 0; i < len; i++) b.put((uint8_t)text[i], 8);
    b.align();                                      /* rbsp_trailing_bits */
    std::vector<uint8_t> nal;
    serialize(nal, 39, 1, false, b.out);            /* (never the first unit of its list: a start code of three bytes, nal.cpp:110-118) */
    if (nal.size() > cap) return 0;
    memcpy(out, nal.data(), nal.size());
    return nal.size();
}

/* An SEI unit with one message (SEI::writeSEImessages, sei.cpp:39-73): payload type and size in their 0xff-escaped form, the payload, rbsp trailing bits; prefix (NAL type 39) or
 * suffix (40), behind a start code of three bytes (it is never the first unit of an access unit).  Returns the unit's size, 0 when it does not fit. */
extern "C" size_t x265amd_write_sei(int suffix, int payload_type, const uint8_t* payload, size_t n, uint8_t* out, size_t cap)
{
    if (!out || (n && !payload) || payload_type < 0) return 0;
    Bits b;
    int t = payload_type;
    for (; t >= 0xff; t -= 0xff) b.put(0xff, 8);
    b.put((uint32_t)t, 8);
    size_t size = n;
    for (; size >= 0xff; size -= 0xff) b.put(0xff, 8);
    b.put((uint32_t)size, 8);
    for (size_t i = 0; i < n; i++) b.put(payload[i], 8);
    b.align();
    std::vector<uint8_t> nal;
    serialize(nal, suffix ? 40 : 39, 1, false, b.out);
    if (nal.size() > cap) return 0;
    memcpy(out, nal.data(), nal.size());
    return nal.size();
}
/* the access unit delimiter (Entropy::codeAUD, entropy.cpp:570-591): pic_type 0 / 1 / 2 for I / P / B, behind a start code of four bytes (it opens its access unit) */
extern "C" size_t x265amd_write_aud(int slice_type, uint8_t* out, size_t cap)
{
    if (!out) return 0;
    Bits b;
    b.put((uint32_t)(slice_type == 2 ? 0 : slice_type == 1 ? 1 : 2), 3);       /* x265amd_slice_info.slice_type: 0 B, 1 P, 2 I */
    b.align();
    std::vector<uint8_t> nal;
    serialize(nal, 35, 1, true, b.out);
    if (nal.size() > cap) return 0;
    memcpy(out, nal.data(), nal.size());
    return nal.size();
}

/* Encoder::getStreamHeaders (encoder.cpp:3234-3259): VPS, SPS and PPS NAL units, each behind a 4-byte start code.
 * Entropy::codeVPS / codeSPS / codeVUI / codePPS (entropy.cpp:233-378, :431-502); no scaling lists, no SPS reference picture sets, no HRD. */
extern "C" size_t x265amd_write_stream_headers(const x265amd_stream_params* p, uint8_t* out, size_t cap)
{
    if (!p || p->max_temporal_sub_layers < 1 || p->max_temporal_sub_layers > 7) return 0;
    const int layers = p->max_temporal_sub_layers;
    std::vector<uint8_t> nal;
    {
        Bits b;                                                  /* codeVPS */
        b.put(0, 4); b.put(3, 2); b.put(0, 6);
        b.put((uint32_t)(layers - 1), 3); b.flag(layers == 1); b.put(0xffff, 16);
        profileTier(b, p);
        b.flag(true);
        for (int i = 0; i < layers; i++) { b.ue((uint32_t)(p->max_dec_pic_buffering[i] - 1)); b.ue((uint32_t)p->num_reorder_pics[i]); b.ue((uint32_t)(p->max_latency_increase[i] + 1)); }
        b.put(0, 6); b.ue(0); b.flag(false); b.flag(false);
        b.align();
        serialize(nal, 32, 1, true, b.out);
    }
    {
        Bits b;                                                  /* codeSPS */
        b.put(0, 4); b.put((uint32_t)(layers - 1), 3); b.flag(layers == 1);
        profileTier(b, p);
        b.ue(0);
        b.ue((uint32_t)p->chroma_format_idc);
        if (p->chroma_format_idc == 3) b.flag(false);
        b.ue((uint32_t)p->pic_width); b.ue((uint32_t)p->pic_height);
        b.flag(p->conformance_window != 0);
        if (p->conformance_window)
        {
            const int hs = p->chroma_format_idc == 1 || p->chroma_format_idc == 2, vs = p->chroma_format_idc == 1;
            b.ue((uint32_t)(p->conf_win_offsets[0] >> hs)); b.ue((uint32_t)(p->conf_win_offsets[1] >> hs));
            b.ue((uint32_t)(p->conf_win_offsets[2] >> vs)); b.ue((uint32_t)(p->conf_win_offsets[3] >> vs));
        }
        b.ue((uint32_t)(p->bit_depth - 8)); b.ue((uint32_t)(p->bit_depth - 8));
        b.ue((uint32_t)(p->log2_max_poc_lsb - 4));
        b.flag(true);
        for (int i = 0; i < layers; i++) { b.ue((uint32_t)(p->max_dec_pic_buffering[i] - 1)); b.ue((uint32_t)p->num_reorder_pics[i]); b.ue((uint32_t)(p->max_latency_increase[i] + 1)); }
        b.ue((uint32_t)(p->log2_min_cu_size - 3)); b.ue((uint32_t)p->log2_diff_max_min_cu_size);
        b.ue((uint32_t)(p->tu_log2_min - 2)); b.ue((uint32_t)(p->tu_log2_max - p->tu_log2_min));
        b.ue((uint32_t)(p->tu_max_depth_inter - 1)); b.ue((uint32_t)(p->tu_max_depth_intra - 1));
        b.flag(false);                                           /* scaling_list_enabled_flag */
        b.flag(p->amp != 0); b.flag(p->sao != 0);
        b.flag(false);                                           /* pcm_enabled_flag */
        b.ue(0);                                                 /* num_short_term_ref_pic_sets */
        b.flag(false);                                           /* long_term_ref_pics_present_flag */
        b.flag(p->temporal_mvp != 0); b.flag(p->strong_intra_smoothing != 0);
        b.flag(true);                                            /* vui_parameters_present_flag; codeVUI */
        b.flag(p->aspect_ratio_idc != 0);
        if (p->aspect_ratio_idc)
        {
            b.put((uint32_t)p->aspect_ratio_idc, 8);
            if (p->aspect_ratio_idc == 255) { b.put((uint32_t)p->sar_width, 16); b.put((uint32_t)p->sar_height, 16); }
        }
        b.flag(p->overscan_info_present != 0);
        if (p->overscan_info_present) b.flag(p->overscan_appropriate != 0);
        b.flag(p->video_signal_type_present != 0);
        if (p->video_signal_type_present)
        {
            b.put((uint32_t)p->video_format, 3); b.flag(p->video_full_range != 0); b.flag(p->colour_description_present != 0);
            if (p->colour_description_present) { b.put((uint32_t)p->colour_primaries, 8); b.put((uint32_t)p->transfer_characteristics, 8); b.put((uint32_t)p->matrix_coefficients, 8); }
        }
        b.flag(p->chroma_loc_info_present != 0);
        if (p->chroma_loc_info_present) { b.ue((uint32_t)p->chroma_sample_loc_top); b.ue((uint32_t)p->chroma_sample_loc_bottom); }
        b.flag(false);                                           /* neutral_chroma_indication_flag */
        b.flag(p->field_seq != 0); b.flag(p->frame_field_info_present != 0);
        b.flag(p->default_display_window != 0);
        if (p->default_display_window) for (int i = 0; i < 4; i++) b.ue((uint32_t)p->def_disp_win_offsets[i]);
        b.flag(p->emit_timing_info != 0);
        if (p->emit_timing_info) { b.put(p->num_units_in_tick, 32); b.put(p->time_scale, 32); b.flag(false); }
        b.flag(false);                                           /* vui_hrd_parameters_present_flag */
        b.flag(false);                                           /* bitstream_restriction_flag */
        b.flag(false);                                           /* sps_extension_flag */
        b.align();
        serialize(nal, 33, 1, true, b.out);
    }
    {
        Bits b;                                                  /* codePPS */
        b.ue(0); b.ue(0); b.flag(false); b.flag(false); b.put(0, 3);
        b.flag(p->sign_hide != 0); b.flag(false);
        b.ue((uint32_t)(p->num_ref_idx_default[0] - 1)); b.ue((uint32_t)(p->num_ref_idx_default[1] - 1));
        b.se(p->init_qp_minus26);
        b.flag(p->constrained_intra_pred != 0); b.flag(p->transform_skip != 0);
        b.flag(p->use_dqp != 0);
        if (p->use_dqp) b.ue((uint32_t)p->max_cu_dqp_depth);
        b.se(p->cb_qp_offset); b.se(p->cr_qp_offset); b.flag(p->slice_chroma_qp_offsets_present != 0);
        b.flag(p->weighted_pred != 0); b.flag(p->weighted_bipred != 0); b.flag(p->transquant_bypass != 0);
        b.flag(false);                                           /* tiles_enabled_flag */
        b.flag(p->wpp != 0); b.flag(p->loop_filter_across_slices != 0);
        b.flag(p->deblocking_filter_control_present != 0);
        if (p->deblocking_filter_control_present)
        {
            b.flag(false); b.flag(p->pic_disable_deblocking != 0);
            if (!p->pic_disable_deblocking) { b.se(p->beta_offset_div2); b.se(p->tc_offset_div2); }
        }
        b.flag(false); b.flag(false); b.ue(0); b.flag(false); b.flag(false);
        b.align();
        serialize(nal, 34, 1, true, b.out);
    }
    if (out && cap >= nal.size()) memcpy(out, nal.data(), nal.size());
    return nal.size();
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
    if ((sliceType == 1 && h->weighted_pred) || (sliceType == 0 && h->weighted_bipred))
    {
        /* pred_weight_table() as Entropy::codePredWeightTable writes it (entropy.cpp:1358-1429): the denominators once, then per list the luma flags, the chroma flags and the
         * weights of the references that carry some (chroma offsets as differences from the prediction 128 - ((128 w) >> denom)) */
        b.ue((uint32_t)h->luma_log2_weight_denom); b.se(h->chroma_log2_weight_denom - h->luma_log2_weight_denom);
        for (int list = 0; list < (sliceType == 0 ? 2 : 1); list++)
        {
            for (int r = 0; r < h->num_ref_idx[list]; r++) b.flag(h->wp[list][r][0].present != 0);
            for (int r = 0; r < h->num_ref_idx[list]; r++) b.flag(h->wp[list][r][1].present != 0);
            for (int r = 0; r < h->num_ref_idx[list]; r++)
            {
                const x265amd_weight* w = h->wp[list][r];
                if (w[0].present) { b.se(w[0].w - (1 << w[0].denom)); b.se(w[0].o); }
                if (w[1].present)
                    for (int plane = 1; plane < 3; plane++)
                    {
                        b.se(w[plane].w - (1 << w[1].denom));
                        const int pred = 128 - ((128 * w[plane].w) >> w[plane].denom);
                        b.se(w[plane].o - pred);
                    }
            }
        }
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
    std::vector<uint8_t> nal;
    serialize(nal, h->nal_unit_type, h->temporal_id_plus1 ? h->temporal_id_plus1 : 1, h->first_in_access_unit != 0, b.out);
    nal.insert(nal.end(), data.begin(), data.end());
    if (!nal.back()) nal.push_back(3);
    if (out && cap >= nal.size()) memcpy(out, nal.data(), nal.size());
    return nal.size();
}
