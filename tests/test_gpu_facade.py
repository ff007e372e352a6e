"""The C++ host facade (FMI / BWT / RankArray / RunBuffer, native file I/O) and the bwt_merge CLI,
on the GPU, against the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "bwt-merge_amd", "csrc", "host")
CHARS = np.frombuffer(b"$ACGTN", dtype=np.uint8)


def build_host():
    subprocess.check_call(["make", "-C", HOST, "-s"])


def test_facade_builds(bwtm):
    """CPU check: the facade and the CLI compile and link against the C-ABI library."""
    bwtm.build()
    build_host()
    assert os.path.exists(os.path.join(HOST, "bwt_merge"))
    out = subprocess.run([os.path.join(HOST, "bwt_merge")], capture_output=True, text=True)
    assert out.returncode == 0 and "Usage: bwt_merge [options] input1 input2 [input3 ...] output" in out.stderr


def test_facade_logic_without_gpu(bwtm):
    """CPU check: the facade's codecs and its queries on both forms of the samples (full arrays / the compact form a merge
    downloads), expansion and native round trip (csrc/host/host_cpu_test.cpp)."""
    build_host()
    out = subprocess.run([os.path.join(HOST, "host_cpu_test")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def write_plain(path, fmi):
    CHARS[fmi.symbols].tofile(path)


@pytest.mark.gpu
def test_facade_api_on_gpu(bwtm, oracle, tmp_path):
    build_host()
    ta = oracle.generate_reads(1001, 700, 60); tb = oracle.generate_reads(1002, 500, 80)
    a = oracle.FMI.from_text(ta); b = oracle.FMI.from_text(tb)
    write_plain(tmp_path / "a.plain", a); write_plain(tmp_path / "b.plain", b)
    m, _ = oracle.merge(a.clone(), b.clone(), threads=2)
    m.data.tofile(tmp_path / "expected.bin")
    out = subprocess.run([os.path.join(HOST, "host_api_test"), str(tmp_path / "a.plain"), str(tmp_path / "b.plain"),
                          str(tmp_path / "expected.bin"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cli_chained_merge_with_verification(bwtm, oracle, tmp_path):
    build_host()
    sets = [oracle.generate_reads(4000 + k, 300 + 50 * k, 100 if k != 2 else 150) for k in range(3)]
    names = []
    for k, t in enumerate(sets):
        names.append(str(tmp_path / ("in%d.plain" % k)))
        write_plain(names[-1], oracle.FMI.from_text(t))
    # patterns: substrings of the reads plus some that do not occur
    rng = np.random.default_rng(9)
    pats = []
    for _ in range(200):
        t = sets[int(rng.integers(0, 3))]
        p = int(rng.integers(0, t.size - 20))
        s = t[p:p + int(rng.integers(3, 16))]
        if np.all(s != 0):
            pats.append(CHARS[s].tobytes().decode())
    pats += ["NNNNNNNN", "ACGTACGTACGTACGTACGT"]
    (tmp_path / "patterns.txt").write_text("\n".join(pats) + "\n")
    exe = os.path.join(HOST, "bwt_merge")
    out = subprocess.run([exe, "-i", "plain_default", "-o", "plain_default", "-v", str(tmp_path / "patterns.txt"), "-t", "3", "-s", "7",
                          names[0], names[1], names[2], str(tmp_path / "out.plain")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    text = out.stdout
    # stdout shape of the reference tool
    assert text.startswith("BWT-merge\n\nInput:            ")
    for line in ("Run buffers:      128 MB", "Thread buffers:   256 MB", "Merge buffers:    6", "Threads:          3", "Sequence blocks:  7",
                 "Temp directory:   .", "Read %d patterns of total length %d" % (len(pats), sum(len(p) for p in pats)), "Verification successful",
                 "Total time:       ", "Peak memory:      "):
        assert line in text, line
    assert text.count("BWTs merged in ") == 2
    direct = oracle.FMI.from_text(np.concatenate(sets))
    got = np.fromfile(tmp_path / "out.plain", dtype=np.uint8)
    assert np.array_equal(got, CHARS[direct.symbols])
    # native output, then native input again: same bytes after a no-op round trip through the CLI's reader
    out2 = subprocess.run([exe, "-i", "plain_default", names[0], names[1], str(tmp_path / "m01.native")], capture_output=True, text=True)
    assert out2.returncode == 0, out2.stdout + out2.stderr
    out3 = subprocess.run([exe, "-i", "native,plain_default", "-o", "plain_default", str(tmp_path / "m01.native"), names[2], str(tmp_path / "out2.plain")],
                          capture_output=True, text=True)
    assert out3.returncode == 0, out3.stdout + out3.stderr
    assert np.array_equal(np.fromfile(tmp_path / "out2.plain", dtype=np.uint8), CHARS[direct.symbols])


@pytest.mark.gpu
def test_cli_says_once_when_the_pool_runs_out_of_addresses(bwtm, oracle, tmp_path):
    """A process that has used up the pool's address ranges (here: a 96-MiB limit and every buffer on the mapped path) keeps merging
    correctly from hipMalloc blocks, and the facade says so ONCE on stderr (how many merges a process can run)."""
    build_host()
    sets = [oracle.generate_reads(4200 + k, 6000, 100) for k in range(4)]
    names = []
    for k, t in enumerate(sets):
        names.append(str(tmp_path / ("in%d.plain" % k)))
        write_plain(names[-1], oracle.FMI.from_text(t))
    env = dict(os.environ, BWTM_POOL_VMM_CHUNK="2097152", BWTM_POOL_VMM_MIN="65536", BWTM_POOL_VA_SEGMENT="33554432", BWTM_POOL_VA_LIMIT="100663296")
    out = subprocess.run([os.path.join(HOST, "bwt_merge"), "-i", "plain_default", "-o", "plain_default"] + names + [str(tmp_path / "out.plain")],
                         capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stderr.count("has used up its address ranges") == 1, out.stderr[-2000:]
    direct = oracle.FMI.from_text(np.concatenate(sets))
    assert np.array_equal(np.fromfile(tmp_path / "out.plain", dtype=np.uint8), CHARS[direct.symbols])


@pytest.mark.gpu
def test_cli_one_thread_per_gpu(bwtm, oracle, tmp_path):
    """bwt_merge -g 0,0,0: three host threads, each with its own context (here on the same GPU), shard the search and
    produce one output slice each; the file equals the single-GPU result, also for a chained merge."""
    build_host()
    sets = [oracle.generate_reads(4100 + k, 2500 + 300 * k, 100) for k in range(3)]
    names = []
    for k, t in enumerate(sets):
        names.append(str(tmp_path / ("in%d.plain" % k)))
        write_plain(names[-1], oracle.FMI.from_text(t))
    exe = os.path.join(HOST, "bwt_merge")
    outs = {}
    # several GPUs: partitioned records by default, sequence blocks with -B; "rounds": the sharded upload of the sequence-block path in three
    # rounds of pieces (what inputs of 128 MiB per GPU and more take by themselves)
    for label, g, extra, env in (("one", "0", [], {}), ("three", "0,0,0", [], {}), ("four", "0,0,0,0", [], {}), ("five_strict", "0,0,0,0,0", ["-P"], {}),
                                 ("blocks3", "0,0,0", ["-B"], {}), ("blocks4", "0,0,0,0", ["-B"], {}), ("rounds", "0,0,0", ["-B"], {"BWTM_SHARDED_UPLOAD_ROUNDS": "3"})):
        out = subprocess.run([exe, "-g", g] + extra + ["-i", "plain_default", names[0], names[1], names[2], str(tmp_path / (label + ".native"))],
                             capture_output=True, text=True, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.count("BWTs merged in ") == 2
        if g != "0":
            mode = "sequence blocks" if "-B" in extra else "partitioned records"
            assert out.stderr.count("mergeMultiGPU(): %d GPUs (%s): upload " % (len(g.split(",")), mode)) == 2, out.stderr[-1500:]     # the phases of both merges
        outs[label] = np.fromfile(tmp_path / (label + ".native"), dtype=np.uint8)
    for label in outs:
        assert np.array_equal(outs["one"], outs[label]), label
    # a part that runs out of room: the default mode repeats the merge with sequence blocks (and says so), -P fails
    small = dict(os.environ, BWTM_TUNE="part_capacity=500")
    out = subprocess.run([exe, "-g", "0,0,0", "-i", "plain_default", names[0], names[1], names[2], str(tmp_path / "fallback.native")], capture_output=True, text=True, env=small)
    assert out.returncode == 0 and out.stderr.count("repeating it with sequence blocks") >= 1, out.stderr[-1500:]
    assert np.array_equal(outs["one"], np.fromfile(tmp_path / "fallback.native", dtype=np.uint8))
    out = subprocess.run([exe, "-g", "0,0,0", "-P", "-i", "plain_default", names[0], names[1], names[2], str(tmp_path / "strict.native")], capture_output=True, text=True, env=small)
    assert out.returncode != 0 and "the merge over partitioned records failed" in out.stderr
    direct = oracle.FMI.from_text(np.concatenate(sets))
    assert np.array_equal(outs["three"][32:32 + direct.nbytes], direct.data)           # header (24 B) + byte count (8 B), then BWT::data


@pytest.mark.gpu
def test_cli_two_distinct_gpus(bwtm, oracle, tmp_path):
    """bwt_merge -g 0,1: two host threads on two DEVICES, the rank-array shards combined by ncclAllReduce (the branch that
    contexts of one GPU never take).  Runs wherever two GPUs are visible; the file must equal the single-GPU one."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    build_host()
    sets = [oracle.generate_reads(4200 + k, 3000 + 500 * k, 100) for k in range(3)]
    names = []
    for k, t in enumerate(sets):
        names.append(str(tmp_path / ("in%d.plain" % k)))
        write_plain(names[-1], oracle.FMI.from_text(t))
    exe = os.path.join(HOST, "bwt_merge")
    outs = {}
    for label, g in (("one", "0"), ("two", "0,1")):
        out = subprocess.run([exe, "-g", g, "-i", "plain_default", names[0], names[1], names[2], str(tmp_path / (label + ".native"))],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        outs[label] = np.fromfile(tmp_path / (label + ".native"), dtype=np.uint8)
    assert np.array_equal(outs["one"], outs["two"])


@pytest.mark.gpu
def test_cli_errors_like_reference(bwtm, tmp_path):
    build_host()
    exe = os.path.join(HOST, "bwt_merge")
    out = subprocess.run([exe, "-i", "nonsense", "a", "b", "c"], capture_output=True, text=True)
    assert out.returncode != 0 and "bwt_merge: Invalid input format: nonsense" in out.stderr
    out = subprocess.run([exe, "a", "b"], capture_output=True, text=True)
    assert out.returncode != 0 and "bwt_merge: Output file not specified" in out.stderr
    out = subprocess.run([exe, str(tmp_path / "missing1"), str(tmp_path / "missing2"), str(tmp_path / "out")], capture_output=True, text=True)
    assert out.returncode != 0 and "Cannot open input file" in out.stderr


@pytest.mark.gpu
def test_cli_ingest_reads_then_merge(bwtm, oracle, tmp_path):
    """bwt_ingest: reads (text, one per line, ragged) -> BWT; against the oracle's brute-force BWT, and the merge of two
    ingested halves against the ingest of the whole."""
    build_host()
    rng = np.random.default_rng(11)
    reads = []
    for k in range(900):
        n = int(rng.integers(0, 140)) if k % 9 else 100
        reads.append(bytes(CHARS[1 + rng.integers(0, 5, size=n)]).decode())
    reads[5] = reads[4]; reads[17] = "acgtnacgtX"                       # a duplicate; lower case and a foreign character
    halves = (reads[:400], reads[400:])
    for name, part in (("a", halves[0]), ("b", halves[1]), ("all", reads)):
        (tmp_path / (name + ".txt")).write_text("\n".join(part) + "\n")
    ingest = os.path.join(HOST, "bwt_ingest")
    for name, leaf in (("a", 150), ("b", 524288), ("all", 256)):
        out = subprocess.run([ingest, "-l", str(leaf), str(tmp_path / (name + ".txt")), str(tmp_path / (name + ".bwt"))], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "Read %d reads of total length %d" % (len(open(tmp_path / (name + ".txt")).read().splitlines()),
                                                      sum(len(r) for r in (halves[0] if name == "a" else halves[1] if name == "b" else reads))) in out.stdout
    out = subprocess.run([os.path.join(HOST, "bwt_merge"), str(tmp_path / "a.bwt"), str(tmp_path / "b.bwt"), str(tmp_path / "ab.bwt")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert (tmp_path / "ab.bwt").read_bytes() == (tmp_path / "all.bwt").read_bytes()
    # the oracle's BWT of the same collection, through the plain format
    out = subprocess.run([os.path.join(HOST, "bwt_convert"), "-i", "native", "-o", "plain_default", str(tmp_path / "all.bwt"), str(tmp_path / "all.plain")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    comp = np.full(256, 5, dtype=np.uint8)
    for ch, c in zip(b"ACGTacgt", [1, 2, 3, 4, 1, 2, 3, 4]):
        comp[ch] = c
    text = np.concatenate([np.concatenate([comp[np.frombuffer(r.encode(), dtype=np.uint8)], np.zeros(1, dtype=np.uint8)]) for r in reads])
    ref = oracle.FMI.from_text(text)
    assert (tmp_path / "all.plain").read_bytes() == CHARS[ref.symbols].tobytes()
