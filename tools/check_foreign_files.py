#!/usr/bin/env python3
"""One command that retires the three structural caveats the day a foreign file is available (README "what is NOT claimed"):

    python tools/check_foreign_files.py DIR [--formats native,sdsl,rfm,ropebwt,sga]

For every regular file in DIR, and for every format it might be in (guessed from its extension and magic bytes: *.native / NativeHeader tag ->
native, *.rope / *.fmr / "RLE\\6" -> ropebwt, *.sga / *.bwt with the SGA magic -> sga, *.rfm, *.sdsl; --formats overrides the guess):

  1. load it through the C++ facade and write it back in the SAME format (bwt_convert -i F -o F): the bytes must be identical -- that is
     BWT::load / BWT::serialize (bwt.cpp:111-148), CumulativeArray::load (support.cpp:442-464), the SDSL framing of csrc/host/sdsl_compat.h,
     and the RopeBWT / SGA readers (formats.cpp:281-445) against a file that the repo's own writer did NOT produce;
  2. convert it to plain_default and back into F and compare again (the payload survives the repo's own codecs);
  3. print what bwt_inspect says about it.

Exit code 0 = every file round-tripped in at least one format; the table says which.  Needs no GPU (bwt_convert and bwt_inspect are host code).
Files the repo's own writers produced prove nothing here; the tests already cover those (tests/test_formats_host.py).
"""
import argparse
import filecmp
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "bwt-merge_amd", "csrc", "host")
ALL = ["native", "sdsl", "rfm", "ropebwt", "sga", "plain_default", "plain_sorted"]


def guess(path):
    ext = os.path.splitext(path)[1].lower().lstrip(".")
    with open(path, "rb") as f:
        head = f.read(8)
    out = []
    if head[:4] == bytes.fromhex("21425754"):                       # NativeHeader::DEFAULT_TAG 0x54574221, little-endian (formats.h:35)
        out.append("native")
    if head[:4] == b"RLE\x06":                                      # RopeHeader tag (formats.cpp:246)
        out.append("ropebwt")
    if head[:2] == bytes.fromhex("ca5b"):                           # SGAHeader tag 0xCACA low bytes differ by writer; the reader decides
        out.append("sga")
    by_ext = {"native": "native", "bwt": "sga", "sga": "sga", "rope": "ropebwt", "fmr": "ropebwt", "rfm": "rfm", "sdsl": "sdsl", "plain": "plain_default"}
    if ext in by_ext and by_ext[ext] not in out:
        out.append(by_ext[ext])
    return out or ["native", "sdsl", "rfm", "ropebwt", "sga"]


def run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True)
    return p.returncode, (p.stdout + p.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--formats", default="")
    args = ap.parse_args()
    convert, inspect = os.path.join(HOST, "bwt_convert"), os.path.join(HOST, "bwt_inspect")
    if not (os.path.exists(convert) and os.path.exists(inspect)):
        subprocess.check_call(["make", "-C", HOST, "-s", "bwt_convert", "bwt_inspect"])
    files = sorted(f for f in (os.path.join(args.directory, n) for n in os.listdir(args.directory)) if os.path.isfile(f))
    if not files:
        print("no files in %s" % args.directory)
        return 2
    print("| file | format | reloads and re-serialises to the same bytes | survives plain_default and back | bwt_inspect |\n|---|---|---|---|---|")
    all_ok = True
    with tempfile.TemporaryDirectory() as tmp:
        for path in files:
            formats = [f for f in args.formats.split(",") if f] or guess(path)
            file_ok = False
            rc_i, text_i = run([inspect, path])
            summary = " / ".join(l.strip() for l in text_i.splitlines() if l.strip())[:160]
            for fmt in formats:
                if fmt not in ALL:
                    print("| %s | %s | unknown format | | |" % (os.path.basename(path), fmt)); continue
                again, plain, back = (os.path.join(tmp, n) for n in ("again", "plain", "back"))
                rc1, out1 = run([convert, "-i", fmt, "-o", fmt, path, again])
                same1 = (rc1 == 0 and filecmp.cmp(path, again, shallow=False))
                rc2, _ = run([convert, "-i", fmt, "-o", "plain_default", path, plain]) if rc1 == 0 else (1, "")
                rc3, _ = run([convert, "-i", "plain_default", "-o", fmt, plain, back]) if rc2 == 0 else (1, "")
                same2 = (rc3 == 0 and filecmp.cmp(path, back, shallow=False))
                why = "" if rc1 == 0 else " (" + (out1.strip().splitlines() or ["failed"])[-1][:80] + ")"
                print("| %s | %s | %s%s | %s | %s |" % (os.path.basename(path), fmt, same1, why, same2 if rc1 == 0 else "-", summary))
                file_ok = file_ok or same1
            all_ok = all_ok and file_ok
    print("\nevery file round-tripped byte for byte in at least one format: %s" % all_ok)
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
