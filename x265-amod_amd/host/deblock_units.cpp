/* Coding data of a picture in the form the deblocking kernels read (include/x265amd.h: x265amd_deblock_units): host C++.
 *
 * The reference's Deblock::deblockCU walks the CU / TU / PU structure of every CTU and marks the 4-sample edges to filter
 * (reference: source/common/deblock.cpp:70-185: setEdgefilterTU over the transform tree, setEdgefilterPU per part size, the CU border),
 * then derives the boundary strength from prediction modes, coded block flags, reference pictures and motion vectors (:201-266).
 * Here the same marks are derived per 4x4 unit from the picture-wide maps the analysis fills in. */
#include "x265amd.h"
#include <string.h>

extern "C" int x265amd_deblock_units(const x265amd_slice_info* si, const x265amd_mvpred_info* info, const x265amd_cu_unit* units, const x265amd_mv_unit* motion,
                                     x265amd_deblock_unit* out)
{
    if (!si) return X265AMD_EINVAL;
    return x265amd_deblock_units_rows(si, info, units, motion, out, 0, si->pic_height >> 2);
}

extern "C" int x265amd_deblock_units_rows(const x265amd_slice_info* si, const x265amd_mvpred_info* info, const x265amd_cu_unit* units, const x265amd_mv_unit* motion,
                                          x265amd_deblock_unit* out, int y4_begin, int y4_end)
{
    if (!si) return X265AMD_EINVAL;
    return x265amd_deblock_units_rect(si, info, units, motion, out, y4_begin, y4_end, 0, si->pic_width >> 2);
}

extern "C" int x265amd_deblock_units_rect(const x265amd_slice_info* si, const x265amd_mvpred_info* info, const x265amd_cu_unit* units, const x265amd_mv_unit* motion,
                                          x265amd_deblock_unit* out, int y4_begin, int y4_end, int x4_begin, int x4_end)
{
    if (!si || !info || !units || !motion || !out || y4_begin < 0 || y4_end > (si->pic_height >> 2) || x4_begin < 0 || x4_end > (si->pic_width >> 2)) return X265AMD_EINVAL;
    const int w4 = si->pic_width >> 2;
    /* picture identities: equal POC = same picture, whatever the list */
    int pocs[32], npoc = 0;
    auto ident = [&](int list, int refIdx) -> int {
        if (refIdx < 0) return -1;
        const int poc = info->ref_poc[list][refIdx];
        for (int i = 0; i < npoc; i++) if (pocs[i] == poc) return i;
        pocs[npoc] = poc;
        return npoc++;
    };
    /* the same numbers whichever rows a call covers: every reference picture of the slice, list 0 first */
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < info->num_ref_idx[l] && r < 16; r++) (void)ident(l, r);
    for (int y4 = y4_begin; y4 < y4_end; y4++)
        for (int x4 = x4_begin; x4 < x4_end; x4++)
        {
            const x265amd_cu_unit& u = units[y4 * w4 + x4];
            const x265amd_mv_unit& m = motion[y4 * w4 + x4];
            x265amd_deblock_unit& d = out[y4 * w4 + x4];
            memset(&d, 0, sizeof(d));
            const int x = x4 * 4, y = y4 * 4;
            const int cuSize = 64 >> u.depth, cuX = x & ~(cuSize - 1), cuY = y & ~(cuSize - 1);
            const int tuSize = cuSize >> u.tu_depth;
            const bool intra = u.pred_mode == X265AMD_MODE_INTRA;
            int flags = 0;
            if (intra) flags |= X265AMD_DB_INTRA;
            if ((u.cbf[0] >> u.tu_depth) & 1) flags |= X265AMD_DB_CBF;
            if (u.tq_bypass) flags |= X265AMD_DB_BYPASS;
            if (!(x & (tuSize - 1))) flags |= X265AMD_DB_TU_LEFT;
            if (!(y & (tuSize - 1))) flags |= X265AMD_DB_TU_TOP;
            const int rx = x - cuX, ry = y - cuY, q = cuSize >> 2, h = cuSize >> 1;
            switch (u.part_size)
            {
            case 1: if (ry == h) flags |= X265AMD_DB_PU_TOP; break;                    /* 2NxN */
            case 2: if (rx == h) flags |= X265AMD_DB_PU_LEFT; break;                   /* Nx2N */
            case 3: if (ry == h) flags |= X265AMD_DB_PU_TOP; if (rx == h) flags |= X265AMD_DB_PU_LEFT; break;
            case 4: if (ry == q) flags |= X265AMD_DB_PU_TOP; break;                    /* 2NxnU */
            case 5: if (ry == 3 * q) flags |= X265AMD_DB_PU_TOP; break;                /* 2NxnD */
            case 6: if (rx == q) flags |= X265AMD_DB_PU_LEFT; break;                   /* nLx2N */
            case 7: if (rx == 3 * q) flags |= X265AMD_DB_PU_LEFT; break;               /* nRx2N */
            default: break;
            }
            d.flags = (uint8_t)flags;
            d.qp = u.qp;
            for (int l = 0; l < 2; l++)
            {
                const bool used = !intra && (m.inter_dir & (1 << l)) && m.ref_idx[l] >= 0;
                d.ref[l] = (int8_t)(used ? ident(l, m.ref_idx[l]) : -1);
                d.mv[l][0] = used ? m.mv[l][0] : 0; d.mv[l][1] = used ? m.mv[l][1] : 0;
                if (npoc > 31) return X265AMD_EINVAL;
            }
        }
    return X265AMD_OK;
}
