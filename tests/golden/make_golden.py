#!/usr/bin/env python3
"""Generate the golden vectors for the hot-path primitives FROM THE REFERENCE ITSELF.

Runs every case of tests/hevc_testlib.py through oracle/_ref/librefprims{8,10}.so (the reference's own C primitive
table, built from /root/reference by oracle/build_ref.sh) and writes
  tests/golden/prims_digests.json   sha256 of the outputs of every (case, depth, mode, rep)
  tests/golden/prims_<depth>.npz    full output arrays for (case, "random", rep 0)
Inputs are not stored: they are regenerated from the per-case seed (hevc_testlib.case_seed).
Only runs in the build container (needs oracle/_ref); the outputs are committed.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hevc_testlib as T

REPS = {"random": 3, "min": 1, "max": 1}


def main():
    digests = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        full = {}
        for name in sorted(T.CASES):
            for mode in T.MODES:
                for rep in range(REPS[mode]):
                    outs = T.run_case(ref, name, mode, rep)
                    digests["%s/%d/%s/%d" % (name, depth, mode, rep)] = T.digest(outs)
                    if mode == "random" and rep == 0:
                        for i, a in enumerate(outs):
                            full["%s/%04d" % (name, i)] = a
        np.savez_compressed(os.path.join(T.GOLDEN_DIR, "prims_%d.npz" % depth), **full)
    with open(os.path.join(T.GOLDEN_DIR, "prims_digests.json"), "w") as f:
        json.dump({"reference": "DJATOM/x265-aMod 3.6+1-aa7f602f7 [noasm] C primitives", "digests": digests}, f, indent=0, sort_keys=True)
    print("wrote", len(digests), "digests")
    make_me_golden()
    make_tu_golden()
    make_intra_golden()
    make_mc_golden()
    make_entropy_golden()
    make_inter_cost_golden()
    make_intra_tu_golden()
    make_inter_search_golden()
    make_inter_rd_golden()
    make_ctu_analysis_golden()
    make_intra_rd_golden()
    make_frame_pipeline_golden()


ME_CONFIGS = [(T.ME_HEX, 2), (T.ME_HEX, 0), (T.ME_HEX, 1), (T.ME_HEX, 5), (T.ME_HEX, 7), (T.ME_DIA, 0), (T.ME_DIA, 2),
              (T.ME_STAR, 2), (T.ME_STAR, 4)]
ME_SCENES = ((1, (5, -3)), (2, (-17, 9)), (3, (0, 0)), (4, (33, 21)))
ME_CHROMA_CONFIGS = [(T.ME_STAR, 3), (T.ME_HEX, 3), (T.ME_STAR, 4), (T.ME_HEX, 7), (T.ME_DIA, 5)]
ME_CHROMA_SCENES = ((11, (6, -4)), (12, (-18, 10)))


def make_me_golden():
    """Results of the reference's own MotionEstimate::motionEstimate (via ref_motion_estimate) and sha256 of its
    BitCost tables -> tests/golden/me_golden.npz"""
    import ctypes as C
    import hashlib
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        for method, subme in ME_CONFIGS:
            for seed, motion in ME_SCENES:
                cur, rp, stride, origin = T.me_make_planes(depth, seed, motion=motion)
                jobs = T.me_jobs(seed * 100 + method * 10 + subme, 60, motion=motion, methods=(method,), submes=(subme,))
                out["me/%d/%d/%d/%d" % (depth, method, subme, seed)] = T.me_run_host(ref, cur, rp, stride, origin, jobs)
        for method, subme in ME_CHROMA_CONFIGS:
            for seed, motion in ME_CHROMA_SCENES:
                cur, rp, stride, cstride, origin, corg = T.me_make_yuv(depth, seed, motion=motion)
                jobs = T.me_jobs(seed * 100 + method * 10 + subme, 50, motion=motion, methods=(method,), submes=(subme,))
                out["mec/%d/%d/%d/%d" % (depth, method, subme, seed)] = T.me_run_host_c(ref, cur, rp, stride, cstride, origin, corg, jobs)
        ref.lib.ref_mvcost_table.restype = C.POINTER(C.c_uint16)
        dig = []
        for qp in range(70):
            p = ref.lib.ref_mvcost_table(qp)
            a = np.ctypeslib.as_array(C.cast(C.addressof(p.contents) - 2 * 65536, C.POINTER(C.c_uint16)), (2 * 65536 + 1,))
            dig.append(np.frombuffer(hashlib.sha256(a.tobytes()).digest(), np.uint8))
        out["mvcost_sha256/%d" % depth] = np.stack(dig)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "me_golden.npz"), **out)
    print("wrote me_golden.npz with", len(out), "arrays")



def make_tu_golden():
    """Quant::transformNxN / invtransformNxN results and RDCost values from the reference build -> tests/golden/tu_golden.npz"""
    import ctypes as C
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        for seed in range(4):
            cases = T.tu_cases(depth, 100 + seed, 150)
            res = T.tu_run_host(ref, cases)
            out["tu/%d/%d/numsig" % (depth, seed)] = np.array([r[0] for r in res], np.int32)
            out["tu/%d/%d/coeff" % (depth, seed)] = np.concatenate([r[1] for r in res])
            out["tu/%d/%d/resi" % (depth, seed)] = np.concatenate([r[2].ravel() for r in res])
        rng = np.random.default_rng(5)
        rows = []
        for _ in range(300):
            qp, st = int(rng.integers(0, 70)), int(rng.integers(0, 3))
            psy = float(rng.choice([0.0, 1.0, 2.0, 0.7]))
            dist, bits, pc = int(rng.integers(0, 1 << 24)), int(rng.integers(0, 1 << 16)), int(rng.integers(0, 1 << 16))
            a = np.zeros(6, np.uint64)
            ref.lib.ref_rdcost(qp, st, C.c_double(psy), C.c_uint64(dist), C.c_uint32(bits), C.c_uint32(pc), T._ptr(a))
            rows.append(a)
        out["rdcost/%d" % depth] = np.stack(rows)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "tu_golden.npz"), **out)
    print("wrote tu_golden.npz with", len(out), "arrays")


def make_intra_golden():
    """neighbour sets of the reference's Predict::initAdiPattern and the 35-mode sa8d scan -> tests/golden/intra_golden.npz"""
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        for seed in range(4):
            res = T.intra_run_host(ref, T.intra_cases(depth, 500 + seed, 200))
            out["intra/%d/%d/ref" % (depth, seed)] = np.concatenate([r[0] for r in res])
            out["intra/%d/%d/flt" % (depth, seed)] = np.concatenate([r[1] for r in res if r[1] is not None])
            out["intra/%d/%d/sa8d" % (depth, seed)] = np.stack([r[2] for r in res])
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "intra_golden.npz"), **out)
    print("wrote intra_golden.npz with", len(out), "arrays")


def make_mc_golden():
    """outputs of the reference's own Predict::motionCompensation -> sha256 per scene in tests/golden/mc_golden.npz"""
    import hashlib
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        for seed in range(3):
            pics, stride, cstride, org = T.mc_make_refs(depth, 900 + seed)
            res = T.mc_run_host(ref, pics, stride, cstride, org, T.mc_jobs(900 + seed, 400))
            out["mc/%d/%d" % (depth, seed)] = np.frombuffer(T.mc_digest(res), np.uint8)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "mc_golden.npz"), **out)
    print("wrote mc_golden.npz with", len(out), "arrays")


def make_entropy_golden():
    """context initialisation, estBit tables, RDOQ levels and bits-only coefficient coding of the reference build
    -> tests/golden/entropy_golden.npz"""
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        out["reset/%d" % depth] = np.stack([T.entropy_reset(ref, st, qp) for st in range(3) for qp in range(52)])
        ests = []
        for st in range(3):
            for qp in (0, 17, 30, 43, 51):
                ctx = T.entropy_reset(ref, st, qp)
                for log2 in range(2, 6):
                    for luma in (1, 0):
                        if luma or log2 < 5:
                            ests.append(T.est_bit(ref, ctx, log2, luma))
        out["est/%d" % depth] = np.stack(ests)
        for seed in range(3):
            cases = T.rdoq_cases(depth, 700 + seed, 250, ctxlib=ref)
            res = T.rdoq_run(ref, cases)
            bits = T.coeff_bits_run(ref, cases, res)
            out["rdoq/%d/%d/numsig" % (depth, seed)] = np.array([r[0] for r in res], np.int32)
            out["rdoq/%d/%d/coeff" % (depth, seed)] = np.concatenate([r[1] for r in res])
            out["rdoq/%d/%d/bits" % (depth, seed)] = np.array([b[0] for b in bits], np.uint64)
            out["rdoq/%d/%d/ctx" % (depth, seed)] = np.stack([b[1] for b in bits])
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "entropy_golden.npz"), **out)
    print("wrote entropy_golden.npz with", len(out), "arrays")


def make_inter_cost_golden():
    """costs of inter prediction candidates from the reference's Predict + primitives -> tests/golden/inter_cost_golden.npz"""
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        for seed in range(2):
            pics, stride, cstride, org = T.mc_make_refs(depth, 1300 + seed, nref=4)
            cost, _ = T.inter_cost_run_host(ref, pics[:3], pics[3], stride, cstride, org, T.inter_cost_jobs(40 + seed, 500))
            out["cost/%d/%d" % (depth, seed)] = cost
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "inter_cost_golden.npz"), **out)
    print("wrote inter_cost_golden.npz with", len(out), "arrays")


def make_intra_tu_golden():
    """fused intra TU step with the reference's Predict / Quant classes -> sha256 per case set in tests/golden/intra_tu_golden.npz"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tit", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_intra_tu.py"))
    tit = importlib.util.module_from_spec(spec); spec.loader.exec_module(tit)
    out = {}
    for depth in (8, 10):
        ref = T.load_ref(depth)
        for seed in range(2):
            cases = T.intra_tu_cases(depth, 2100 + seed, 300, rdoq=bool(seed))
            out["digest/%d/%d" % (depth, seed)] = tit.digest(T.intra_tu_run_host(ref, cases))
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "intra_tu_golden.npz"), **out)
    print("wrote intra_tu_golden.npz with", len(out), "arrays")


def make_inter_search_golden():
    """results of the reference's Search::predInterSearch on CUData / Slice / MotionReference fixtures -> tests/golden/inter_search_golden.npz"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tis", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_inter_search.py"))
    tis = importlib.util.module_from_spec(spec); spec.loader.exec_module(tis)
    out = {}
    for i, (depth, seed, b) in enumerate(tis.CASES):
        bits, pus, dig = tis.pack(T.inter_search_run_ref(T.load_ref(depth), T.inter_search_case(depth, seed, b)))
        out["bits/%d" % i], out["pus/%d" % i], out["pred/%d" % i] = bits, pus, dig
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "inter_search_golden.npz"), **out)
    print("wrote inter_search_golden.npz with", len(out), "arrays")


def make_inter_rd_golden():
    """results of the reference's Search::encodeResAndCalcRdInterCU on CUData / Slice / Search fixtures -> tests/golden/inter_rd_golden.npz"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tird", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_inter_rd.py"))
    tird = importlib.util.module_from_spec(spec); spec.loader.exec_module(tird)
    out = {}
    for k, (depth, seed, st, td, psy) in enumerate(tird.CASES):
        c = T.rd_case(depth, seed, st, td, psy)
        for i, d in enumerate(T.rd_pack(T.rd_run_ref(T.load_ref(depth), c), c)):
            for name, a in d.items():
                out["%d/%d/%s" % (k, i, name)] = a
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "inter_rd_golden.npz"), **out)
    print("wrote inter_rd_golden.npz with", len(out), "arrays")
    out = {}
    for k, (depth, seed, st, psy) in enumerate(tird.SKIP_CASES):
        c = T.skip_case(depth, seed, st, psy)
        for i, d in enumerate(T.rd_pack(T.skip_run_ref(T.load_ref(depth), c), c)):
            for name, a in d.items():
                out["%d/%d/%s" % (k, i, name)] = a
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "skip_rd_golden.npz"), **out)
    print("wrote skip_rd_golden.npz with", len(out), "arrays")


def make_ctu_analysis_golden():
    """results of the reference's Analysis::compressCTU on fixtures -> tests/golden/ctu_analysis_golden.npz"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tca", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_ctu_analysis.py"))
    tca = importlib.util.module_from_spec(spec); spec.loader.exec_module(tca)
    out = {}
    for k, cfg in enumerate(tca.CASES + tca.PART_CASES + tca.RDOQ_CASES):
        c = tca.make_case(k)
        for i, d in enumerate(T.ctu_pack(T.ctu_run_ref(T.load_ref(cfg[0]), c))):
            for name, a in d.items():
                out["%d/%d/%s" % (k, i, name)] = a
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "ctu_analysis_golden.npz"), **out)
    print("wrote ctu_analysis_golden.npz with", len(out), "arrays")


def make_intra_rd_golden():
    """results of the reference's Search::checkIntraInInter + encodeIntraInInter on fixtures -> tests/golden/intra_rd_golden.npz"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tir", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_intra_rd.py"))
    tir = importlib.util.module_from_spec(spec); spec.loader.exec_module(tir)
    out = {}
    for k, (depth, seed, st, psy, strong) in enumerate(tir.CASES):
        c = T.intra_rd_case(depth, seed, st, psy, strong=strong)
        for i, d in enumerate(T.intra_rd_pack(T.intra_rd_run_ref(T.load_ref(depth), c), c)):
            for name, a in d.items():
                out["%d/%d/%s" % (k, i, name)] = a
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "intra_rd_golden.npz"), **out)
    print("wrote intra_rd_golden.npz with", len(out), "arrays")
    out = {}
    for k, cfg in enumerate(tir.CHECK_CASES):
        depth, seed, st, psy, strong = cfg[:5]
        c = T.check_intra_case(depth, seed, st, psy, strong=strong, tu_intra=cfg[5] if len(cfg) > 5 else 0)
        for i, d in enumerate(T.intra_rd_pack(T.check_intra_run_ref(T.load_ref(depth), c), c)):
            for name, a in d.items():
                out["%d/%d/%s" % (k, i, name)] = a
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "check_intra_golden.npz"), **out)
    print("wrote check_intra_golden.npz with", len(out), "arrays")


def make_frame_pipeline_golden():
    """the reference ENCODER's own output for the clip of T.frame_clip(): reconstructed frames and slice payloads -> frame_pipeline_golden.npz"""
    import subprocess, tempfile, csv
    frames, stride, cstride, org = T.frame_clip(8, 4)
    out = {"nframes": np.array(4)}
    for tag, cli in (("", T.FRAME_CLI_ARGS), ("deblock/", [a for a in T.FRAME_CLI_ARGS if a != "--no-deblock"]),
                     ("wpp/", [("2" if a == "none" else a) for a in T.FRAME_CLI_ARGS if a not in ("--no-deblock", "--no-wpp")] + ["--wpp"])):   # WPP needs a thread pool
        _frame_pipeline_one(frames, stride, cstride, org, tag, cli, out)
    # a clip with B frames: fixed mini-GOP (I P b b P b b in coding order), deblocking on
    framesb, stride, cstride, org = T.frame_clip_b(8)
    clib = [a for a in T.FRAME_CLI_ARGS if a != "--no-deblock"]
    clib[clib.index("--bframes") + 1] = "2"
    clib[clib.index("--rc-lookahead") + 1] = "5"
    _frame_pipeline_one(framesb, stride, cstride, org, "bframes/", clib + ["--no-b-pyramid"], out, nframes=7)
    _frame_pipeline_one(framesb, stride, cstride, org, "sao_bframes/", [a for a in clib if a != "--no-sao"] + ["--no-b-pyramid", "--sao"], out, nframes=7)
    _frame_pipeline_one(frames, stride, cstride, org, "sao/", [a for a in T.FRAME_CLI_ARGS if a not in ("--no-deblock", "--no-sao")] + ["--sao"], out)
    # rectangular + asymmetric partitions (with and without --limit-modes) on both clips, deblocking on
    _frame_pipeline_one(framesb, stride, cstride, org, "rectamp_bframes/", clib + ["--no-b-pyramid", "--rect", "--amp"], out, nframes=7)
    _frame_pipeline_one(frames, stride, cstride, org, "rectamp_lm/", [a for a in T.FRAME_CLI_ARGS if a != "--no-deblock"] + ["--rect", "--amp", "--limit-modes"], out)
    # rd 5 (RD on every candidate: compressInterCU_rd5_6) with and without rect / amp
    _frame_pipeline_one(framesb, stride, cstride, org, "rd5_bframes/", [("5" if (a == "3" and clib[i - 1] == "--rd") else a) for i, a in enumerate(clib)] + ["--no-b-pyramid"], out, nframes=7)
    cli5 = [a for a in T.FRAME_CLI_ARGS if a != "--no-deblock"]
    cli5[cli5.index("--rd") + 1] = "6"
    _frame_pipeline_one(frames, stride, cstride, org, "rd6_rectamp/", cli5 + ["--rect", "--amp", "--limit-modes"], out)
    # rd 2 (SA8D mode choice, only the winner coded)
    cli2 = list(clib)
    cli2[cli2.index("--rd") + 1] = "2"
    _frame_pipeline_one(framesb, stride, cstride, org, "rd2_bframes/", cli2 + ["--no-b-pyramid"], out, nframes=7)
    cli2 = [a for a in T.FRAME_CLI_ARGS if a != "--no-deblock"]
    cli2[cli2.index("--rd") + 1] = "2"
    _frame_pipeline_one(frames, stride, cstride, org, "rd2_rectamp/", cli2 + ["--rect", "--amp", "--limit-modes"], out)
    # mini-GOP / keyframe logic of the encoder object (tests/test_encoder_api.py): 3 B frames with a short last mini-GOP; IDR every 4 frames
    cli3 = list(clib)
    cli3[cli3.index("--bframes") + 1] = "3"
    _frame_pipeline_one(framesb, stride, cstride, org, "bframes3/", cli3 + ["--no-b-pyramid"], out, nframes=7)
    clik = list(clib)
    clik[clik.index("--keyint") + 1] = "4"
    _frame_pipeline_one(framesb, stride, cstride, org, "keyint/", clik + ["--no-b-pyramid", "--min-keyint", "4"], out, nframes=7)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "frame_pipeline_golden.npz"), **out)
    print("wrote frame_pipeline_golden.npz:", [[len(out[t + "slice/%d" % i]) for i in range(4)] for t in ("", "deblock/", "wpp/")])


def _frame_pipeline_one(frames, stride, cstride, org, tag, cli, out, nframes=4):
    import subprocess, tempfile, csv
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "clip.y4m"), "wb") as f:
            f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420\n" % (T.MC_W, T.MC_H))
            for p in frames:
                f.write(b"FRAME\n")
                for pl in T.frame_planes(p, stride, cstride, org):
                    f.write(np.ascontiguousarray(pl).tobytes())
        exe = os.path.join(T.REF_DIR, "x265_ref8")
        r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv", "--csv", "log.csv", "--csv-log-level", "1"] + cli,
                           cwd=d, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
        hevc = open(os.path.join(d, "out.hevc"), "rb").read()
        qps, sched = [], []
        for row in csv.reader(open(os.path.join(d, "log.csv"))):
            if len(row) > 8 and row[1].strip().endswith("SLICE"):
                qps.append(int(float(row[3])))
                l0 = [int(v) for v in row[7].split()] if row[7].strip() != "-" else []
                l1 = [int(v) for v in row[8].split()] if row[8].strip() != "-" else []
                t = row[1].strip()
                # type: 2 I, 1 P, 0 B; referenced: upper-case type names
                sched.append([{"I": 2, "P": 1, "B": 0, "b": 0}[t[0]], int(row[2]), int(t[0] != "b")] + (l0 + [-1] * 4)[:4] + (l1 + [-1] * 4)[:4])
    out[tag + "slice_qp"] = np.array(qps, np.int32)
    out[tag + "schedule"] = np.array(sched, np.int32)          # per coded frame: type, poc, referenced, L0 pocs (4), L1 pocs (4)
    fsz = T.MC_W * T.MC_H * 3 // 2
    for k in range(nframes):
        fr = rec[k * fsz:(k + 1) * fsz]                    # output (POC) order
        out[tag + "recon/%d/0" % k] = fr[:T.MC_W * T.MC_H].reshape(T.MC_H, T.MC_W)
        out[tag + "recon/%d/1" % k] = fr[T.MC_W * T.MC_H:T.MC_W * T.MC_H * 5 // 4].reshape(T.MC_H // 2, T.MC_W // 2)
        out[tag + "recon/%d/2" % k] = fr[T.MC_W * T.MC_H * 5 // 4:].reshape(T.MC_H // 2, T.MC_W // 2)
    # NAL units (Annex B): VCL units (types 0..31) in order; payload with emulation prevention bytes removed, 2-byte NAL header dropped
    pos, nals = 0, []
    starts = []
    i = 0
    while i + 3 <= len(hevc):
        if hevc[i:i + 3] == b"\x00\x00\x01":
            starts.append(i + 3); i += 3
        else:
            i += 1
    for a, b in zip(starts, starts[1:] + [len(hevc) + 4]):
        end = b - 3 if b <= len(hevc) else len(hevc)
        nal = hevc[a:end]
        while nal.endswith(b"\x00") and b <= len(hevc):
            nal = nal[:-1]
        nals.append(nal)
    out[tag + "stream"] = np.frombuffer(hevc, np.uint8)
    k = 0
    for nal in nals:
        if ((nal[0] >> 1) & 0x3f) < 32:
            out[tag + "nal/%d" % k] = np.frombuffer(nal, np.uint8)          # as in the byte stream (escaped), without the start code
            rbsp = bytearray(); z = 0
            for byte in nal[2:]:
                if z >= 2 and byte == 3:
                    z = 0
                    continue
                rbsp.append(byte)
                z = z + 1 if byte == 0 else 0
            out[tag + "slice/%d" % k] = np.frombuffer(bytes(rbsp), np.uint8)
            k += 1
    assert k == nframes, k


def make_lowres_golden():
    """Lowres::init planes (digests) and LookaheadTLD::lowresIntraEstimate outputs of the reference -> tests/golden/lowres_golden.npz"""
    import importlib.util, hashlib
    spec = importlib.util.spec_from_file_location("tl", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_lowres.py"))
    tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)
    out = {}
    for k, (depth, seed, crop) in enumerate(tl.CASES):
        c = T.lowres_case(depth, seed, crop)
        planes, cost, mode, rows, lc, sums = T.lowres_run_ref(T.load_ref(depth), c)
        out["%d/plane_md5" % k] = np.array([hashlib.md5(np.ascontiguousarray(p).tobytes()).hexdigest() for p in planes])
        out["%d/cost" % k], out["%d/mode" % k], out["%d/row_satds" % k], out["%d/lowres_costs" % k], out["%d/sums" % k] = cost, mode, rows, lc, sums
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "lowres_golden.npz"), **out)
    print("wrote lowres_golden.npz with", len(out), "arrays")


def make_lowres_cost_golden():
    """CostEstimateGroup::estimateFrameCost of the reference on Lowres fixtures -> tests/golden/lowres_cost_golden.npz"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tl", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_lowres.py"))
    tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)
    out = {}
    for k, (depth, seed, crop, p0, b, p1) in enumerate(tl.COST_CASES):
        c = T.lowres_cost_case(depth, seed, crop)
        r = T.lowres_cost_run_ref(T.load_ref(depth), c, p0, b, p1)
        for name, a in r.items():
            out["%d/%s" % (k, name)] = a
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "lowres_cost_golden.npz"), **out)
    print("wrote lowres_cost_golden.npz with", len(out), "arrays")


def make_aq_energy_golden():
    """LookaheadTLD::acEnergyCu of the reference over whole pictures -> tests/golden/aq_energy_golden.npz"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tl", os.path.join(os.path.dirname(T.GOLDEN_DIR), "test_lowres.py"))
    tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)
    out = {}
    for k, (depth, seed, W, H, qg) in enumerate(tl.AQ_CASES):
        out["%d/energy" % k], out["%d/wp" % k] = T.aq_run_ref(T.load_ref(depth), T.aq_case(depth, seed), W, H, qg)
    for k, (ci, mode, strength, bias) in enumerate(tl.AQ_OFFSET_CASES):
        depth, seed, W, H, qg = tl.AQ_CASES[ci]
        out["o%d/qp_aq_offset" % k], out["o%d/qp_cutree_offset" % k], out["o%d/inv_qscale_factor" % k] = T.aq_offsets_ref(T.load_ref(depth), T.aq_case(depth, seed), W, H, mode, strength, bias, qg)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "aq_energy_golden.npz"), **out)
    print("wrote aq_energy_golden.npz with", len(out), "arrays")


def make_encoder_api_golden():
    """whole streams + per-frame reconstruction digests of the reference encoder for clips the frame-pipeline goldens do not cover:
    picture sizes that are not multiples of the CTU size (partial CTUs at the right / bottom edge) and a 14-frame clip (the DPB evicts
    pictures) -> tests/golden/encoder_api_golden.npz"""
    import subprocess, tempfile, hashlib
    out = {}
    for tag, (w, h), nframes, extra in (("crop_p/", (200, 152), 4, ["--bframes", "0"]),
                                        ("crop_b/", (248, 184), 7, ["--bframes", "2", "--no-b-pyramid", "--sao", "--wpp", "--pools", "2"]),
                                        ("long/", (128, 128), 14, ["--bframes", "2", "--no-b-pyramid", "--ref", "4"]),
                                        ("hbd_b/", (192, 136), 7, ["--bframes", "2", "--no-b-pyramid", "--sao", "--rect", "--amp"]),
                                        ("hbd_rd5/", (128, 128), 4, ["--bframes", "0", "--rd", "5"]),
                                        ("wvga/", (832, 480), 5, ["--bframes", "2", "--no-b-pyramid", "--sao", "--wpp", "--pools", "4"]),
                                        # the option space of the built subset (tests/test_encoder_api.py OPTION_CONFIGS)
                                        ("opt_a/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--me", "star", "--subme", "3", "--ref", "2", "--max-merge", "5"]),
                                        ("opt_b/", (192, 128), 5, ["--bframes", "0", "--me", "dia", "--subme", "1", "--ref", "1", "--max-merge", "2", "--no-early-skip", "--rskip", "0"]),
                                        ("opt_c/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--tu-inter-depth", "3", "--tu-intra-depth", "3", "--limit-refs", "0", "--no-signhide"]),
                                        ("opt_d/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--no-strong-intra-smoothing", "--no-temporal-mvp", "--no-b-intra", "--limit-refs", "1"]),
                                        ("opt_e/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--subme", "5", "--rd", "4", "--sao"]),
                                        ("opt_f/", (192, 128), 4, ["--bframes", "0", "--qp", "22", "--subme", "4"]),
                                        ("opt_g/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--qp", "38", "--subme", "7", "--me", "star", "--rd", "5", "--rect", "--amp"]),
                                        ("opt_h/", (192, 128), 4, ["--bframes", "0", "--subme", "0", "--rd", "2", "--tu-inter-depth", "2"]),
                                        # rate-distortion optimised quantisation (the slow presets): level 1, level 2 with psy-rdoq, with TU splits
                                        ("preset_veryfast/", (192, 128), 10, ["--preset", "veryfast"]),
                                        ("preset_fast/", (192, 128), 10, ["--preset", "fast"]),
                                        ("preset_slow/", (192, 128), 10, ["--preset", "slow"]),
                                        ("preset_veryslow/", (192, 128), 10, ["--preset", "veryslow"]),
                                        ("placebo_notskip/", (192, 128), 10, ["--preset", "placebo", "--no-tskip"]),
                                        ("hbd_slow/", (192, 128), 6, ["--preset", "slow"]),
                                        ("hbd_veryslow/", (192, 128), 6, ["--preset", "veryslow", "--bframes", "3"]),
                                        ("opt_j/", (192, 128), 4, ["--bframes", "0", "--qp", "10"]),
                                        ("opt_k/", (192, 128), 6, ["--bframes", "3", "--no-b-pyramid", "--qp", "45"]),
                                        ("opt_l/", (192, 128), 10, ["--bframes", "2", "--no-b-pyramid", "--ref", "6", "--max-merge", "1"]),
                                        ("opt_m/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--no-deblock", "--sao"]),      # (with --wpp on top the reference program itself hangs)
                                        ("opt_n/", (192, 128), 10, ["--bframes", "1", "--no-b-pyramid", "--keyint", "3", "--min-keyint", "3"]),
                                        ("opt_o/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--rd", "4", "--rect", "--limit-modes", "--limit-refs", "2", "--subme", "6", "--me", "dia"]),
                                        ("opt_p/", (328, 248), 5, ["--preset", "slow", "--wpp", "--pools", "4", "--bframes", "2"]),
                                        ("hbd_wpp/", (328, 248), 5, ["--bframes", "2", "--no-b-pyramid", "--sao", "--wpp", "--pools", "4", "--rect", "--amp"]),
                                        ("opt_q/", (192, 128), 4, ["--bframes", "0", "--qp", "48"]),
                                        ("opt_r/", (192, 128), 8, ["--bframes", "4", "--no-b-pyramid", "--ref", "1", "--no-early-skip", "--rd", "5", "--rect"]),
                                        ("opt_s/", (640, 368), 5, ["--bframes", "2", "--no-b-pyramid", "--me", "star", "--merange", "24", "--subme", "7", "--max-merge", "4", "--rect", "--amp",
                                                                   "--sao", "--wpp", "--pools", "4"]),
                                        ("fhd/", (1920, 1080), 4, ["--bframes", "2", "--no-b-pyramid", "--sao", "--wpp", "--pools", "8"]),       # BASELINE.json configs[1] geometry
                                        ("rdoq_a/", (192, 128), 4, ["--bframes", "0", "--rdoq-level", "1"]),
                                        ("rdoq_b/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--rdoq-level", "2", "--psy-rdoq", "1.0", "--rd", "4"]),
                                        ("rdoq_c/", (192, 128), 5, ["--bframes", "2", "--no-b-pyramid", "--rdoq-level", "2", "--psy-rdoq", "2.5", "--tu-inter-depth", "3",
                                                                    "--tu-intra-depth", "3", "--rd", "5", "--no-signhide"])):
        depth = 10 if tag.startswith("hbd") else 8
        planes = T.encoder_api_clip(tag, w, h, nframes, depth)
        cli = [a for a in T.FRAME_CLI_ARGS if a != "--no-deblock"]
        # later options override earlier ones on the reference's command line
        cli = cli + ["--rc-lookahead", "5"] + extra
        if "--preset" in extra:
            # a whole preset's analysis settings: only what the built subset cannot do yet is switched off (AQ, cutree, weighted prediction, adaptive GOPs, rate control)
            cli = ["--preset", extra[extra.index("--preset") + 1], "--qp", "30", "--aq-mode", "0", "--no-cutree", "--no-weightp", "--no-weightb", "--b-adapt", "0", "--no-scenecut",
                   "--keyint", "250", "--no-wpp", "--frame-threads", "1", "--pools", "none", "--no-info", "--no-open-gop", "--rc-lookahead", "10", "--lookahead-slices", "0",
                   "--no-b-pyramid"] + [a for a in extra if a not in ("--preset", extra[extra.index("--preset") + 1])]
        if "--sao" in extra:
            cli = [a for a in cli if a not in ("--no-sao", "--no-wpp")]
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "clip.y4m"), "wb") as f:
                f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
                for fr in planes:
                    f.write(b"FRAME\n")
                    for pl in fr:
                        f.write(np.ascontiguousarray(pl).tobytes())
            exe = os.path.join(T.REF_DIR, "x265_ref%d" % depth)
            r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv"] + cli, cwd=d, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
            fsz = w * h * 3 // 2 * (2 if depth == 10 else 1)
            assert len(rec) == fsz * nframes
            out[tag + "stream"] = np.frombuffer(open(os.path.join(d, "out.hevc"), "rb").read(), np.uint8)
            out[tag + "recon_md5"] = np.array([hashlib.md5(rec[k * fsz:(k + 1) * fsz].tobytes()).hexdigest() for k in range(nframes)])
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "encoder_api_golden.npz"), **out)
    print("wrote encoder_api_golden.npz:", {k: len(v) for k, v in out.items()})


def make_encoder_ft_golden():
    """streams + reconstruction digests of the reference encoder run with several frame threads (its default on any machine with four cores or more): the
    frame-parallel rules of search.cpp:77-92 / sao.cpp:264 -> tests/golden/encoder_ft_golden.npz.  For the clips whose motion reaches beyond the lag the
    stream must differ from the --frame-threads 1 stream (otherwise the case would not test the rules)."""
    import subprocess, tempfile, hashlib
    out = {}
    for tag, ((w, h), nframes, depth, kind, _, extra) in T.FT_CASES.items():
        planes = T.encoder_ft_frames(tag)
        streams = {}
        for ft in ("3", "1", "2"):
            cli = [a for a in T.FT_CLI]
            cli[cli.index("--frame-threads") + 1] = ft
            cli = cli + extra
            with tempfile.TemporaryDirectory() as d:
                with open(os.path.join(d, "clip.y4m"), "wb") as f:
                    f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
                    for fr in planes:
                        f.write(b"FRAME\n")
                        for pl in fr:
                            f.write(np.ascontiguousarray(pl).tobytes())
                exe = os.path.join(T.REF_DIR, "x265_ref%d" % depth)
                r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv"] + cli, cwd=d, capture_output=True, text=True, timeout=600)
                assert r.returncode == 0, r.stderr[-2000:]
                streams[ft] = open(os.path.join(d, "out.hevc"), "rb").read()
                if ft == "3":
                    rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
                    fsz = w * h * 3 // 2 * (2 if depth == 10 else 1)
                    assert len(rec) == fsz * nframes
                    out[tag + "recon_md5"] = np.array([hashlib.md5(rec[k * fsz:(k + 1) * fsz].tobytes()).hexdigest() for k in range(nframes)])
        assert streams["3"] == streams["2"], tag + ": the frame-parallel stream depends on the thread count"
        print(tag, len(streams["3"]), "bytes; differs from --frame-threads 1:", streams["3"] != streams["1"])
        if kind.startswith("down"):
            assert streams["3"] != streams["1"], tag + ": the clip does not reach beyond the lag"
        out[tag + "stream"] = np.frombuffer(streams["3"], np.uint8)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "encoder_ft_golden.npz"), **out)
    print("wrote encoder_ft_golden.npz")


def make_encoder_preset_golden(only=None):
    """the presets as they come (CRF 28 + aq-mode 2 + cuTree: no --qp) -> tests/golden/encoder_preset_golden.json"""
    make_encoder_full_golden(only, T.PRESET_CASES, T.PRESET_CLI, "encoder_preset_golden.json")


def make_encoder_full_golden(only=None, cases=None, tail=None, name="encoder_full_golden.json"):
    """BASELINE.json's configurations 3-5 at their stated size (and an rd 2 clip of 1080 rows for complexityCheckCU): the reference encoder's stream digest + size and
    the digest of every reconstructed picture -> tests/golden/encoder_full_golden.json.  The clips are SURVEY.md section 8d's generator (hevc_testlib.survey_clip)."""
    import subprocess, tempfile, hashlib, time
    path = os.path.join(T.GOLDEN_DIR, name)
    out = json.load(open(path)) if os.path.exists(path) else {}
    for tag, ((w, h), nframes, depth, cfg_id, _, extra) in (cases or T.FULL_CASES).items():
        if only and tag not in only:
            continue
        planes = T.full_case_frames(tag)
        cli = extra + (tail if tail is not None else T.FULL_CLI)
        with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
            with open(os.path.join(d, "clip.y4m"), "wb") as f:
                f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
                for fr in planes:
                    f.write(b"FRAME\n")
                    for pl in fr:
                        f.write(np.ascontiguousarray(pl).tobytes())
            exe = os.path.join(T.REF_DIR, "x265_ref%d" % depth)
            t0 = time.time()
            r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv"] + cli, cwd=d, capture_output=True, text=True, timeout=7200)
            assert r.returncode == 0, r.stderr[-2000:]
            fsz = w * h * 3 // 2 * (2 if depth == 10 else 1)
            rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
            assert len(rec) == fsz * nframes
            stream = open(os.path.join(d, "out.hevc"), "rb").read()
            out[tag] = {"stream_md5": hashlib.md5(stream).hexdigest(), "stream_bytes": len(stream),
                        "recon_md5": [hashlib.md5(rec[k * fsz:(k + 1) * fsz].tobytes()).hexdigest() for k in range(nframes)],
                        "reference_command_line": " ".join(cli), "reference_seconds": round(time.time() - t0, 1)}
            print(tag, out[tag], r.stderr.strip().splitlines()[-1])
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


def make_encoder_sc_golden():
    """streams + reconstruction digests + frame types of the reference encoder with scene-cut detection on (--scenecut 40, --b-adapt 0) -> tests/golden/encoder_sc_golden.npz"""
    import subprocess, tempfile, hashlib
    out = {}
    for tag, ((w, h), nframes, depth, cuts, _, extra) in T.SC_CASES.items():
        planes = T.scene_case_frames(tag)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "clip.y4m"), "wb") as f:
                f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
                for fr in planes:
                    f.write(b"FRAME\n")
                    for pl in fr:
                        f.write(np.ascontiguousarray(pl).tobytes())
            exe = os.path.join(T.REF_DIR, "x265_ref%d" % depth)
            r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv", "--csv", "log.csv", "--csv-log-level", "1"] + T.SC_CLI + extra, cwd=d, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
            fsz = w * h * 3 // 2 * (2 if depth == 10 else 1)
            assert len(rec) == fsz * nframes
            types = []
            for line in open(os.path.join(d, "log.csv")).read().splitlines()[1:]:
                c = [x.strip() for x in line.split(",")]
                if len(c) > 3 and c[0].isdigit():
                    types.append("%s:%s" % (c[2], c[1].split("-")[0]))
            out[tag + "stream"] = np.frombuffer(open(os.path.join(d, "out.hevc"), "rb").read(), np.uint8)
            out[tag + "recon_md5"] = np.array([hashlib.md5(rec[k * fsz:(k + 1) * fsz].tobytes()).hexdigest() for k in range(nframes)])
            out[tag + "types"] = np.array(types)
            print(tag, len(out[tag + "stream"]), "bytes", " ".join(types))
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "encoder_sc_golden.npz"), **out)


def make_encoder_ba_golden():
    """the same for --b-adapt 2 (T.BA_CASES) -> tests/golden/encoder_ba_golden.npz"""
    import subprocess, tempfile, hashlib
    out = {}
    for tag, ((w, h), nframes, depth, _, _, extra) in T.BA_CASES.items():
        planes = T.ba_case_frames(tag)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "clip.y4m"), "wb") as f:
                f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
                for fr in planes:
                    f.write(b"FRAME\n")
                    for pl in fr:
                        f.write(np.ascontiguousarray(pl).tobytes())
            exe = os.path.join(T.REF_DIR, "x265_ref%d" % depth)
            r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv", "--csv", "log.csv", "--csv-log-level", "1"] + T.BA_CLI + extra, cwd=d, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
            fsz = w * h * 3 // 2 * (2 if depth == 10 else 1)
            assert len(rec) == fsz * nframes
            types = []
            for line in open(os.path.join(d, "log.csv")).read().splitlines()[1:]:
                c = [x.strip() for x in line.split(",")]
                if len(c) > 3 and c[0].isdigit():
                    types.append("%s:%s" % (c[2], c[1].split("-")[0]))
            out[tag + "stream"] = np.frombuffer(open(os.path.join(d, "out.hevc"), "rb").read(), np.uint8)
            out[tag + "recon_md5"] = np.array([hashlib.md5(rec[k * fsz:(k + 1) * fsz].tobytes()).hexdigest() for k in range(nframes)])
            out[tag + "types"] = np.array(types)
            print(tag, len(out[tag + "stream"]), "bytes", " ".join(types))
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "encoder_ba_golden.npz"), **out)


def make_encoder_og_golden(cases=None, frames_of=None, cli=None, name="encoder_og_golden.npz"):
    """the same with open GOPs (T.OG_CASES: CRA keyframes, leading pictures) -> tests/golden/encoder_og_golden.npz; with a B pyramid (T.BP_CASES) -> encoder_bp_golden.npz"""
    import subprocess, tempfile, hashlib
    out = {}
    cases = cases or T.OG_CASES; frames_of = frames_of or T.og_case_frames; cli = cli or T.OG_CLI
    for tag, ((w, h), nframes, depth, _, _, extra) in cases.items():
        planes = frames_of(tag)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "clip.y4m"), "wb") as f:
                f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
                for fr in planes:
                    f.write(b"FRAME\n")
                    for pl in fr:
                        f.write(np.ascontiguousarray(pl).tobytes())
            exe = os.path.join(T.REF_DIR, "x265_ref%d" % depth)
            r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv", "--csv", "log.csv", "--csv-log-level", "1"] + cli + extra, cwd=d, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
            fsz = w * h * 3 // 2 * (2 if depth == 10 else 1)
            assert len(rec) == fsz * nframes
            types = []
            for line in open(os.path.join(d, "log.csv")).read().splitlines()[1:]:
                c = [x.strip() for x in line.split(",")]
                if len(c) > 3 and c[0].isdigit():
                    types.append("%s:%s" % (c[2], c[1].split("-")[0]))
            out[tag + "stream"] = np.frombuffer(open(os.path.join(d, "out.hevc"), "rb").read(), np.uint8)
            out[tag + "recon_md5"] = np.array([hashlib.md5(rec[k * fsz:(k + 1) * fsz].tobytes()).hexdigest() for k in range(nframes)])
            out[tag + "types"] = np.array(types)
            print(tag, len(out[tag + "stream"]), "bytes", " ".join(types))
    if name == "encoder_wp_golden.npz":
        # the fade clip: what the reference's weight analysis logs for its first weighted P picture (--log-level full: "poc: N weights: [L0:R0 Y{scale/2^denom+offset}...")
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "clip.y4m"), "wb") as f:
                f.write(b"YUV4MPEG2 W320 H192 F30:1 Ip A1:1 C420\n")
                for fr in T.wp_fade_frames():
                    f.write(b"FRAME\n")
                    for pl in fr:
                        f.write(np.ascontiguousarray(pl).tobytes())
            r = subprocess.run([os.path.join(T.REF_DIR, "x265_ref8"), "--input", "clip.y4m", "-o", "out.hevc", "--log-level", "full"] + T.WP_CLI + T.WP_FADE_CLI, cwd=d, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            lines = [l.split("]:", 1)[1].strip() for l in r.stderr.splitlines() if "weights:" in l]
            out["wp_fade/weights"] = np.array(lines)
            print("wp_fade/", lines[:3])
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, name), **out)


def make_encoder_fade_golden():
    """the fade clips (hevc_testlib.FADE_CASES) through the reference with --log-level full: the stream, the reconstructions, the frame types, and what its weight analysis logs
    per picture ("poc: N weights: [L0:R0 Y{scale/2^denom+offset}U{...}V{...}] [L1:R0 ...]")"""
    import subprocess, tempfile, hashlib
    out = {}
    for tag, ((w, h), nframes, depth, _, _, extra) in T.FADE_CASES.items():
        planes = T.fade_case_frames(tag)
        with tempfile.TemporaryDirectory() as d:
            with open(os.path.join(d, "clip.y4m"), "wb") as f:
                f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 %s\n" % (w, h, b"C420p10" if depth == 10 else b"C420"))
                for fr in planes:
                    f.write(b"FRAME\n")
                    for pl in fr:
                        f.write(np.ascontiguousarray(pl).tobytes())
            exe = os.path.join(T.REF_DIR, "x265_ref%d" % depth)
            r = subprocess.run([exe, "--input", "clip.y4m", "-o", "out.hevc", "--recon", "rec.yuv", "--csv", "log.csv", "--csv-log-level", "1", "--log-level", "full"] + T.WP_CLI + extra,
                               cwd=d, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            rec = np.fromfile(os.path.join(d, "rec.yuv"), np.uint8)
            fsz = w * h * 3 // 2 * (2 if depth == 10 else 1)
            assert len(rec) == fsz * nframes
            types = []
            for line in open(os.path.join(d, "log.csv")).read().splitlines()[1:]:
                c = [x.strip() for x in line.split(",")]
                if len(c) > 3 and c[0].isdigit():
                    types.append("%s:%s" % (c[2], c[1].split("-")[0]))
            lines = [l.split("]:", 1)[1].strip() for l in r.stderr.splitlines() if "weights:" in l]
            out[tag + "stream"] = np.frombuffer(open(os.path.join(d, "out.hevc"), "rb").read(), np.uint8)
            out[tag + "recon_md5"] = np.array([hashlib.md5(rec[k * fsz:(k + 1) * fsz].tobytes()).hexdigest() for k in range(nframes)])
            out[tag + "types"] = np.array(types)
            out[tag + "weights"] = np.array(lines)
            print(tag, len(out[tag + "stream"]), "bytes", " ".join(types))
            for l in lines:
                print("   ", l)
    np.savez_compressed(os.path.join(T.GOLDEN_DIR, "encoder_fade_golden.npz"), **out)


# ---- rate control (tests/test_ratecontrol.py): cuTree digests from the reference's own Lookahead::cuTree, and records of whole reference encodes ----
RC_ENCODES = [
    ("scene 320x192 x30, the preset as it comes", "scene", (320, 192, 30), [], dict()),
    ("survey 416x240 x40 (a scene change at 24)", "survey", (416, 240, 40), [], dict()),
    ("crf 22, two B pictures", "survey", (416, 240, 30), ["crf=22", "bframes=2"], dict(rf=22.0, bframes=2)),
    ("no B pyramid, keyframes every 12", "api", (320, 192, 30), ["b-pyramid=0", "keyint=12", "min-keyint=12"], dict(keyint=12)),
    ("no B pictures", "survey", (416, 240, 20), ["bframes=0"], dict(bframes=0)),
    ("no cuTree: the blurred-complexity branch", "survey", (416, 240, 30), ["cutree=0", "qcomp=0.7"], dict(cutree=0, qcomp=0.7)),
]


def _rc_clip(kind, w, h, n):
    if kind == "scene":
        return T.scene_clip(w, h, n, [n // 2])
    if kind == "survey":
        return T.survey_clip(w, h, 8, 2, 0, n)
    return T.encoder_api_clip("rc", w, h, n)


def make_ratecontrol_golden():
    import ratecontrol_lib as RL
    R = T.load_ref(8)
    out = {"reference": "DJATOM/x265-aMod 3.6+1-aa7f602f7 [noasm]", "cutree": [], "rc": [], "cu_qp": []}
    for seed, kw in RL.CUTREE_CASES:
        c = RL.cutree_case(seed, **kw)
        tree, prop, recalc = RL.cutree_run_ref(R, c)
        out["cutree"].append(RL.digest(tree, prop, recalc))
    for k, (what, kind, (w, h, n), opts, rc) in enumerate(RC_ENCODES):
        recs, stream = RL.reference_rc_records(_rc_clip(kind, w, h, n), w, h, 8, "medium", opts, "/tmp/rc_golden_%d" % k)
        keep = [dict(poc=r["poc"], type=r["type"], referenced=r["referenced"], slice_qp=r["slice_qp"], scenecut=r["scenecut"], ref_poc=r["ref_poc"], satd=r["satd"],
                     qp_rc_bits=str(int(np.float64(r["qp_rc"]).view(np.uint64)))) for r in recs]
        out["rc"].append(dict(what=what, w=w, h=h, rc=rc, records=keep))
        print(what, len(stream), "bytes;", " ".join("%d:%d/%.3f" % (r["poc"], r["slice_qp"], r["qp_rc"]) for r in recs[:12]))
        if k == 1:
            # the QP of the CUs that carry one: CUs of 64 or 32 samples (the quantisation group's depth and above) with a luma residual
            for r in recs[:14]:
                offs = r["cutree"] if r["referenced"] else r["aq"]
                cus = []
                for y in range(0, h, 32):
                    for x in range(0, w, 32):
                        d = int(r["depth"][y // 4, x // 4])
                        if d > 1 or (d == 0 and (x % 64 or y % 64)) or not r["cbf"][y // 4, x // 4] or x + (64 >> d) > w or y + (64 >> d) > h:
                            continue
                        cus.append((x, y, 64 >> d, int(r["qp"][y // 4, x // 4])))
                if cus:
                    out["cu_qp"].append(dict(poc=r["poc"], w=w, h=h, base_bits=str(int(np.float64(r["qp_rc"]).view(np.uint64))),
                                             offsets_bits=[str(int(v)) for v in offs.view(np.uint64)], cus=cus))
            print("   CUs with a coded QP:", sum(len(c["cus"]) for c in out["cu_qp"]))
    with open(os.path.join(T.GOLDEN_DIR, "ratecontrol_golden.json"), "w") as f:
        json.dump(out, f)


# ---- final entropy coding (tests/test_cabac.py): digests of the reference's own Entropy::encodeCTU ... finishSlice on the seeded pictures ----
def make_cabac_golden():
    import hashlib
    CASES = [(1, 128, 128, 2), (2, 200, 136, 1), (3, 264, 72, 0), (4, 64, 64, 0), (5, 136, 200, 1), (6, 192, 128, 0), (7, 128, 64, 2), (8, 320, 192, 0)]       # tests/test_cabac.py
    out = {}
    for depth in (8, 10):
        R = T.load_ref(depth)
        out[str(depth)] = {}
        for i, (seed, w, h, st) in enumerate(CASES):
            for dense in (False, True):
                res = T.cabac_run_ref(R, T.cabac_case(seed + 100 * dense, w, h, st, dense))
                hh = hashlib.sha256()
                hh.update(res[0].tobytes()); hh.update(res[1].tobytes()); hh.update(res[2].tobytes())
                out[str(depth)]["%d/%d" % (i, int(dense))] = hh.hexdigest()
    with open(os.path.join(T.GOLDEN_DIR, "cabac_golden.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("cabac golden:", sum(len(v) for v in out.values()), "digests")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cabac":
        make_cabac_golden()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rcrecords":
        make_ratecontrol_golden()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fade":
        make_encoder_fade_golden()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "og":
        make_encoder_og_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "b1":
        sel = {k: v for k, v in T.B1_CASES.items() if k != "ba1_medium/"}
        make_encoder_og_golden(sel, T.b1_case_frames, T.B1_CLI, "encoder_b1_golden.npz")
        make_encoder_og_golden({"ba1_medium/": T.B1_CASES["ba1_medium/"]}, T.b1_case_frames, T.WP_CLI, "encoder_b1m_golden.npz")
    elif len(sys.argv) > 1 and sys.argv[1] == "wp":
        make_encoder_og_golden(T.WP_CASES, T.wp_case_frames, T.WP_CLI, "encoder_wp_golden.npz")
    elif len(sys.argv) > 1 and sys.argv[1] == "ls":
        make_encoder_og_golden(T.LS_CASES, T.ls_case_frames, T.LS_CLI, "encoder_ls_golden.npz")
    elif len(sys.argv) > 1 and sys.argv[1] == "bp":
        make_encoder_og_golden(T.BP_CASES, T.bp_case_frames, T.BP_CLI, "encoder_bp_golden.npz")
    elif len(sys.argv) > 1 and sys.argv[1] == "sc":
        make_encoder_sc_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "ba":
        make_encoder_ba_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "ft":
        make_encoder_ft_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "full":
        make_encoder_full_golden(sys.argv[2:] or None)
    elif len(sys.argv) > 1 and sys.argv[1] == "preset":
        make_encoder_preset_golden(sys.argv[2:] or None)
    elif len(sys.argv) > 1 and sys.argv[1] == "cli":
        make_encoder_full_golden(sys.argv[2:] or None, T.CLI_CASES, T.PRESET_CLI, "encoder_cli_golden.json")
    elif len(sys.argv) > 1 and sys.argv[1] == "rc":
        make_encoder_full_golden(sys.argv[2:] or None, T.RC_CASES, T.PRESET_CLI, "encoder_rc_golden.json")
    else:
        main()
