"""data_sets::DataSet options of the C++ host side that shape WHICH sequences reach the hot path and in which order
(SURVEY 8 row f2): --truncate_seq (DataSet.cpp:527-542), --train_fraction (:457-458,518-520), several --train_file's (:481-600),
--shuffle_sequences / --shuffle_fractions (:225-243,416-427).  Host only (`--dump_fractions`, no GPU): the driver's fractions
against the Python mirror (lstm-rnn_amd/fraction.py).  The reference's random streams are boost's, so the shuffles are
checked through the properties the reference's algorithm guarantees, not against a stream."""
import json
import os
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from helpers import net_desc, random_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "lstm-rnn_amd", "currennt_hip")
P, C = 6, 4


def write_nc(path, xs, ts, num_labels, prefix="a"):
    f = netcdf_file(path, "w")
    n = sum(len(x) for x in xs)
    f.createDimension("numSeqs", len(xs)); f.createDimension("numTimesteps", n)
    f.createDimension("inputPattSize", xs[0].shape[1]); f.createDimension("numLabels", num_labels)
    f.createDimension("maxSeqTagLength", 16)
    tags = f.createVariable("seqTags", "c", ("numSeqs", "maxSeqTagLength"))
    for i in range(len(xs)):
        tags[i] = np.array(list(("%s%03d" % (prefix, i)).ljust(16, "\0")), "c")
    v = f.createVariable("seqLengths", "i", ("numSeqs",)); v[:] = np.array([len(x) for x in xs], np.int32)
    v = f.createVariable("targetClasses", "i", ("numTimesteps",)); v[:] = np.concatenate(ts).astype(np.int32)
    v = f.createVariable("inputs", "f", ("numTimesteps", "inputPattSize")); v[:] = np.concatenate(xs).astype(np.float32)
    f.close()


def make_file(tmp_path, name, lens, seed, prefix, inputs=P):
    rng = np.random.RandomState(seed)
    xs = [rng.randn(n, inputs).astype(np.float32) for n in lens]
    ts = [rng.randint(0, C, n).astype(np.int32) for n in lens]
    path = str(tmp_path / name)
    write_nc(path, xs, ts, C, prefix)
    return path, xs, ts


def network_file(tmp_path):
    layers = net_desc(P, [("blstm", 8)], C)
    weights = random_weights(layers, np.random.RandomState(5), 0.3)
    net = str(tmp_path / "network.jsn")
    json.dump({"layers": layers, "weights": {k: {a: np.asarray(b).tolist() for a, b in w.items()} for k, w in weights.items()}}, open(net, "w"))
    return net


def dump(args, epochs=1):
    if not os.path.exists(BIN):
        import __graft_entry__ as ge
        ge.build()
    out = subprocess.run([BIN, "--train", "true", "--dump_fractions", "true", "--dump_epochs", str(epochs)] + args,
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    per_epoch = [[]]
    for l in out.stdout.splitlines():
        if l.startswith("EPOCH") and l != "EPOCH 0":
            per_epoch.append([])
        if l.startswith("FRACTION"):
            kv = dict(p.split("=", 1) for p in l.split()[2:])
            kv["tags"] = kv["tags"].split(","); kv["lens"] = [int(v) for v in kv["lens"].split(",")]; kv["pieces"] = [int(v) for v in kv["pieces"].split(",")]
            per_epoch[-1].append(kv)
    return out.stdout, per_epoch


def check_rows_against_mirror(pkg, rows, xs, ts, info, PS, tag_of):
    """every dumped fraction = the Python packer on the sequences the row names, slot by slot"""
    key = {(tag_of(fi, si), k): i for i, (fi, si, k) in enumerate(info)}
    seen = []
    for kv in rows:
        idx = [key[(t, k)] for t, k in zip(kv["tags"], kv["pieces"])]
        seen += idx
        assert [len(xs[i]) for i in idx] == kv["lens"]
        f = pkg.make_fraction([xs[i] for i in idx], [ts[i] for i in idx], PS)
        assert int(kv["T"]) == f["T"] and int(kv["Tmin"]) == f["Tmin"] and int(kv["seqs"]) == f["numSeqs"]
        assert int(kv["none"]) == int((f["patTypes"] == 0).sum())
        assert abs(float(kv["sum_inputs"]) - float(f["inputs"].astype(np.float64).sum())) < 1e-3
        assert int(kv["sum_targets"]) == int(f["targetClasses"][f["targetClasses"] >= 0].sum())
    return seen


def test_truncated_pieces_rule(pkg):
    """DataSet.cpp:527-542: cut while MORE than 1.5 x trunc remains; the remainder is the last piece (0.5 .. 1.5 x trunc)."""
    tp = pkg.fraction.truncated_pieces
    assert tp(40, 10) == [10, 10, 10, 10]
    assert tp(25, 10) == [10, 15]            # 25 > 15 -> 10; 15 is not > 15 -> the rest
    assert tp(16, 10) == [10, 6]
    assert tp(15, 10) == [15]
    assert tp(31, 10) == [10, 10, 11]
    assert tp(7, 10) == [7] and tp(7, 0) == [7] and tp(0, 10) == []
    assert tp(3, 2) == [3] and tp(4, 2) == [2, 2]       # 1.5 x trunc = 3: 3 is kept whole, 4 is cut


def test_truncate_seq_cpu(pkg, tmp_path):
    """--truncate_seq 10: the pieces are sequences of their own (length sort, packing, totals); a piece carries its
    sequence's tag and its piece number (`originalSeqIdx`, DataSet.cpp:530-531)."""
    lens = (40, 7, 25, 16, 10, 31, 15)
    nc, xs, ts = make_file(tmp_path, "train.nc", lens, 3, "a")
    net = network_file(tmp_path)
    PS = 3
    text, (rows,) = dump(["--train_file", nc, "--network", net, "--parallel_sequences", str(PS), "--truncate_seq", "10"])
    px, pt, info = pkg.fraction.load_sequences([(xs, ts)], truncate_seq=10)
    assert [len(x) for x in px] == [10, 10, 10, 10, 7, 10, 15, 10, 6, 10, 10, 10, 11, 15]
    assert "Sequences:        14" in text and "Sequence lengths: 6..15" in text and "Total timesteps:  %d" % sum(lens) in text
    assert len(rows) == (len(px) + PS - 1) // PS
    seen = check_rows_against_mirror(pkg, rows, px, pt, info, PS, lambda fi, si: "a%03d" % si)
    assert sorted(seen) == list(range(len(px)))                          # every piece exactly once
    flat = [n for kv in rows for n in kv["lens"]]
    assert flat == sorted(flat)                                          # training sets are length-sorted (DataSet.cpp:603-605)
    # without the option: one piece per sequence
    _, (rows0,) = dump(["--train_file", nc, "--network", net, "--parallel_sequences", str(PS)])
    assert sorted(n for kv in rows0 for n in kv["lens"]) == sorted(lens) and all(k == 0 for kv in rows0 for k in kv["pieces"])


def test_train_fraction_cpu(pkg, tmp_path):
    """--train_fraction f: the first max(int(numSeqs * f), 1) sequences of the file (DataSet.cpp:518-520); out of (0, 1]
    is "Invalid fraction" (:457-458)."""
    lens = (11, 5, 9, 3, 14, 7, 8)
    nc, xs, ts = make_file(tmp_path, "train.nc", lens, 4, "a")
    net = network_file(tmp_path)
    for frac, n_expected in ((0.5, 3), (0.01, 1), (1.0, 7), (0.43, 3), (0.86, 6)):
        text, (rows,) = dump(["--train_file", nc, "--network", net, "--parallel_sequences", "2", "--train_fraction", str(frac)])
        px, pt, info = pkg.fraction.load_sequences([(xs, ts)], fraction=frac)
        assert len(px) == n_expected and [i[1] for i in info] == list(range(n_expected))
        assert "Loaded fraction:  %d%%" % int(np.float32(frac) * 100) in text and "Sequences:        %d" % n_expected in text
        assert "Total timesteps:  %d" % sum(lens[:n_expected]) in text
        seen = check_rows_against_mirror(pkg, rows, px, pt, info, 2, lambda fi, si: "a%03d" % si)
        assert sorted(seen) == list(range(n_expected))
    out = subprocess.run([BIN, "--train", "true", "--dump_fractions", "true", "--train_file", nc, "--network", net, "--train_fraction", "0"],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "FAILED: Invalid fraction" in out.stdout


def test_several_train_files_cpu(pkg, tmp_path):
    """--train_file a.nc,b.nc: one sequence list over both files (DataSet.cpp:481-600), sorted as a whole; the fraction option
    and the truncation apply per file; files that do not fit together are refused with the reference's words (:502-515)."""
    nc_a, xa, ta = make_file(tmp_path, "a.nc", (11, 5, 9, 22), 6, "a")
    nc_b, xb, tb = make_file(tmp_path, "b.nc", (4, 13, 8, 6, 30), 7, "b")
    net = network_file(tmp_path)
    tag_of = lambda fi, si: "%s%03d" % ("ab"[fi], si)
    text, (rows,) = dump(["--train_file", nc_a + "," + nc_b, "--network", net, "--parallel_sequences", "4"])
    px, pt, info = pkg.fraction.load_sequences([(xa, ta), (xb, tb)])
    assert "Sequences:        9" in text and "Sequence lengths: 4..30" in text
    seen = check_rows_against_mirror(pkg, rows, px, pt, info, 4, tag_of)
    assert sorted(seen) == list(range(9))
    flat = [n for kv in rows for n in kv["lens"]]
    assert flat == sorted(flat)
    assert {t[0] for kv in rows[:1] for t in kv["tags"]} == {"a", "b"}          # the sort interleaves the files
    # fraction 0.5 and truncation 8, per file: a -> first 2 sequences (11, 5), b -> first 2 (4, 13)
    text, (rows,) = dump(["--train_file", nc_a + ";" + nc_b, "--network", net, "--parallel_sequences", "4", "--train_fraction", "0.5", "--truncate_seq", "8"])
    px, pt, info = pkg.fraction.load_sequences([(xa, ta), (xb, tb)], fraction=0.5, truncate_seq=8)
    assert [len(x) for x in px] == [11, 5, 4, 8, 5]
    seen = check_rows_against_mirror(pkg, rows, px, pt, info, 4, tag_of)
    assert sorted(seen) == list(range(5))
    nc_c, _, _ = make_file(tmp_path, "c.nc", (4, 5), 8, "c", inputs=P + 1)
    out = subprocess.run([BIN, "--train", "true", "--dump_fractions", "true", "--train_file", nc_a + "," + nc_c, "--network", net],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "FAILED: Number of inputs mismatch in NC files" in out.stdout


def flat_tags(rows):
    return [t for kv in rows for t in kv["tags"]]


def test_shuffle_sequences_cpu(pkg, tmp_path):
    """--shuffle_sequences true (DataSet.cpp:225-229,420-421): the sequence list is shuffled at the start of EVERY epoch (the
    shuffles accumulate); every sequence is trained on exactly once per epoch; fractions stay parallel_sequences wide.
    Reproducible per --random_seed."""
    lens = tuple(range(3, 20))                                            # 17 sequences, distinct lengths: the sort is unique
    nc, xs, ts = make_file(tmp_path, "train.nc", lens, 9, "a")
    net = network_file(tmp_path)
    PS = 4
    base = ["--train_file", nc, "--network", net, "--parallel_sequences", str(PS)]
    _, (sorted_rows,) = dump(base)
    sorted_order = flat_tags(sorted_rows)
    assert sorted_order == ["a%03d" % i for i in range(17)]
    px, pt, info = pkg.fraction.load_sequences([(xs, ts)])
    _, epochs = dump(base + ["--shuffle_sequences", "true", "--random_seed", "11"], epochs=3)
    assert len(epochs) == 3
    orders = []
    for rows in epochs:
        assert [int(kv["seqs"]) for kv in rows] == [4, 4, 4, 4, 1]        # still PS-sized, the rest in the last one
        seen = check_rows_against_mirror(pkg, rows, px, pt, info, PS, lambda fi, si: "a%03d" % si)
        assert sorted(seen) == list(range(17))                            # every sequence exactly once
        orders.append(flat_tags(rows))
    assert orders[0] != sorted_order and orders[1] != orders[0] and orders[2] != orders[1]
    _, again = dump(base + ["--shuffle_sequences", "true", "--random_seed", "11"], epochs=3)
    assert [flat_tags(r) for r in again] == orders                        # same seed, same run
    _, other = dump(base + ["--shuffle_sequences", "true", "--random_seed", "12"], epochs=1)
    assert flat_tags(other[0]) != orders[0]


def is_chunk_permutation(new, old, PS):
    """new = the PS-sized chunks of old (the last one may be short) in some order"""
    chunks = [tuple(old[a:a + PS]) for a in range(0, len(old), PS)]
    pos = 0
    left = list(chunks)
    while pos < len(new):
        for c in left:
            if tuple(new[pos:pos + len(c)]) == c:
                left.remove(c); pos += len(c)
                break
        else:
            return False
    return not left


def test_shuffle_fractions_cpu(pkg, tmp_path):
    """--shuffle_fractions true (DataSet.cpp:231-248,422-423): the list is cut into chunks of parallel_sequences sequences and
    the CHUNKS are shuffled -- sequences of similar length stay together.  16 sequences / PS 4: every fraction of every epoch
    is one of the four fractions of the sorted list.  17 sequences: the short chunk moves too and the fractions are re-cut from the
    concatenation (the reference's behaviour: `_makeFractionTask` cuts m_sequences at multiples of PS), so what holds is that an
    epoch's order is a chunk permutation of the previous epoch's order."""
    net = network_file(tmp_path)
    PS = 4
    for n_seq in (16, 17):
        lens = tuple(range(3, 3 + n_seq))
        nc, xs, ts = make_file(tmp_path, "train%d.nc" % n_seq, lens, 10, "a")
        base = ["--train_file", nc, "--network", net, "--parallel_sequences", str(PS)]
        _, (sorted_rows,) = dump(base)
        prev = flat_tags(sorted_rows)
        px, pt, info = pkg.fraction.load_sequences([(xs, ts)])
        _, epochs = dump(base + ["--shuffle_fractions", "true", "--random_seed", "21"], epochs=4)
        moved = False
        for rows in epochs:
            seen = check_rows_against_mirror(pkg, rows, px, pt, info, PS, lambda fi, si: "a%03d" % si)
            assert sorted(seen) == list(range(n_seq))
            cur = flat_tags(rows)
            assert is_chunk_permutation(cur, prev, PS), (cur, prev)
            moved = moved or cur != prev
            if n_seq == 16:
                sorted_sets = [set(kv["tags"]) for kv in sorted_rows]
                assert all(set(kv["tags"]) in sorted_sets for kv in rows) and all(int(kv["seqs"]) == PS for kv in rows)
            prev = cur
        assert moved
    # both options: sequences first, then fractions (DataSet.cpp:420-423) -- still every sequence once per epoch
    _, epochs = dump(base + ["--shuffle_fractions", "true", "--shuffle_sequences", "true", "--random_seed", "21"], epochs=2)
    for rows in epochs:
        assert sorted(flat_tags(rows)) == sorted(flat_tags(sorted_rows))
    # without the options nothing moves between epochs
    _, epochs = dump(["--train_file", nc, "--network", net, "--parallel_sequences", str(PS), "--shuffle_fractions", "false"], epochs=2)
    assert flat_tags(epochs[0]) == flat_tags(epochs[1]) == flat_tags(sorted_rows)
