"""Debugging aid: what the device-run inter paths are worth -- the 60-frame bench clip in CQP with them on / off, and under the preset's rate control.  usage: python dbg/chain_worth.py [frames]"""
import os, subprocess, sys
n = sys.argv[1] if len(sys.argv) > 1 else "60"
for tag, env, mode in (("cqp, chains on", {}, "cqp"), ("cqp, inter chain + fused search off", {"X265AMD_INTER_CHAIN": "0"}, "cqp"), ("crf (host paths)", {}, "crf")):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "preset_timing.py"), n, mode], env=e, capture_output=True, text=True)
    print(tag, ":", [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("encode of")])
