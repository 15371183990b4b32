"""Fraction packer: host-side mirror of data_sets::DataSet::_makeFractionTask
(currennt_lib/src/data_sets/DataSet.cpp:300-414) for in-memory sequences.

A fraction is `parallel_sequences` sequences interleaved time-major: slot (t, i) of sequence i
sits at pattern index t*PS + i; slots past a sequence's end (and columns of missing sequences) are
PATTYPE_NONE with zero inputs and target class -1.
"""
import numpy as np

PATTYPE_NONE, PATTYPE_FIRST, PATTYPE_NORMAL, PATTYPE_LAST = 0, 1, 2, 3


def make_fraction(inputs, targets, parallel_sequences, classification=True, output_size=None,
                  context_left=0, context_right=0, output_lag=0):
    """inputs: list of [len_i][P] float arrays; targets: list of [len_i] int arrays (classification)
    or [len_i][L] float arrays.  Returns the dict layout used by NeuralNetwork.load_sequences.
    context_left/right splice neighbouring frames into each input pattern (edge frames repeated) and
    output_lag delays the targets (class 0 / value 1.0 before the lag), DataSet.cpp:302-305,346-397."""
    PS = int(parallel_sequences)
    if not inputs or len(inputs) > PS:
        raise ValueError("need 1..parallel_sequences sequences")
    lengths = [int(x.shape[0]) for x in inputs]
    T, Tmin = max(lengths), min(lengths)                       # DataSet.cpp:316-319
    P0 = int(inputs[0].shape[1])
    P = P0 * (context_left + context_right + 1)
    x = np.zeros((T, PS, P), np.float32)                       # :330
    pat = np.full((T, PS), PATTYPE_NONE, np.int8)              # :331
    frac = {"T": T, "Tmin": Tmin, "PS": PS, "numSeqs": len(inputs), "seqLengths": lengths}
    if classification:
        tc = np.full((T, PS), -1, np.int32)                    # :333-334
    else:
        L = int(output_size if output_size is not None else targets[0].shape[1])
        tg = np.zeros((T, PS, L), np.float32)
    for i, (xi, ti, n) in enumerate(zip(inputs, targets, lengths)):
        for k, off in enumerate(range(-context_left, context_right + 1)):      # :346-366
            src = np.clip(np.arange(n) + off, 0, n - 1)
            x[:n, i, k * P0:(k + 1) * P0] = np.asarray(xi)[src]
        lag = min(int(output_lag), n)
        if classification:
            tc[:lag, i] = 0                                    # :372-380
            tc[lag:n, i] = np.asarray(ti)[:n - lag]
        else:
            tg[:lag, i, :] = 1.0                               # :383-397
            tg[lag:n, i, :] = np.asarray(ti)[:n - lag]
        pat[:n, i] = PATTYPE_NORMAL                            # :400-409
        pat[0, i] = PATTYPE_FIRST
        if n > 1:
            pat[n - 1, i] = PATTYPE_LAST
    frac["inputs"] = x.reshape(T * PS, P)
    frac["patTypes"] = pat.reshape(T * PS)
    if classification:
        frac["targetClasses"] = tc.reshape(T * PS)
    else:
        frac["targets"] = tg.reshape(T * PS, -1)
    return frac


def make_fractions(inputs, targets, parallel_sequences, sort_by_length=False, **kw):
    """Cut a list of sequences into consecutive fractions (DataSet.cpp:632-668); optionally sort
    ascending by length first as the training set does (DataSet.cpp:603-605)."""
    order = list(range(len(inputs)))
    if sort_by_length:
        order.sort(key=lambda i: inputs[i].shape[0])
    out = []
    for a in range(0, len(order), parallel_sequences):
        idx = order[a:a + parallel_sequences]
        out.append(make_fraction([inputs[i] for i in idx], [targets[i] for i in idx],
                                 parallel_sequences, **kw))
    return out


def real_frames(frac):
    return int((np.asarray(frac["patTypes"]) != PATTYPE_NONE).sum())


def truncated_pieces(length, trunc):
    """--truncate_seq (DataSet.cpp:527-542): pieces of `trunc` steps while MORE than 1.5 * trunc steps remain, the remainder
    (0.5 .. 1.5 * trunc steps) as the last piece; trunc <= 0 keeps the sequence whole."""
    out = []
    length = int(length)
    while length > 0:
        n = min(trunc, length) if (trunc > 0 and length > 1.5 * trunc) else length
        out.append(n)
        length -= n
    return out


def load_sequences(files, fraction=1.0, truncate_seq=0):
    """The sequence list data_sets::DataSet builds from its files (DataSet.cpp:443-606), before the length sort.
    files: list of (inputs, targets) pairs, one per NetCDF file, each a list of per-sequence arrays in file order.
    fraction: --train_fraction etc.: the first max(int(float32(numSeqs) * float32(fraction)), 1) sequences of EVERY file
    (:518-520).  Returns (inputs, targets, info) with info[i] = (file index, sequence index in the file, piece number)."""
    if not (0 < fraction <= 1):
        raise ValueError("Invalid fraction")                                         # :457-458
    xs, ts, info = [], [], []
    for fi, (fx, ft) in enumerate(files):
        n_seq = max(int(np.float32(len(fx)) * np.float32(fraction)), 1)
        for si in range(n_seq):
            pos = 0
            for k, n in enumerate(truncated_pieces(len(fx[si]), truncate_seq)):
                xs.append(np.asarray(fx[si])[pos:pos + n]); ts.append(np.asarray(ft[si])[pos:pos + n]); info.append((fi, si, k))
                pos += n
    return xs, ts, info
