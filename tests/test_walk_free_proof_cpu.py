"""The argument every walk-free decoder rests on, checked as a property on the CPU: for a chunk of `nwords`
dwords, C = the dwords whose barcode field equals the chunk's barcode (and that leave room for a header), succ(c) = c + HW + na(c).
If (1) 2 is in C, (2) every c in C has succ(c) = nwords or succ(c) in C, (3) |C| = nrec, (4) the sizes of C's records add up to
nwords - 2 - then C is exactly the set of record starts the sequential parse finds, whatever the other words happen to hold.  The
chunks below are built to attack it: UMIs, refs and alignment counts that spell the barcode, records cut, counts falsified, words
overwritten at random.  (What the kernels do with C is the -m gpu tests' business; this pins the reasoning they share.)"""
import numpy as np

HW = 3   # dwords of a record header: na, barcode, UMI (4-byte fields)


def candidates(w, bc):
    n = len(w)
    return [i for i in range(2, n) if i + HW <= n and w[i + 1] == bc]


def proof_holds(w, bc, nrec):
    """The four checks as the decoders make them (csrc/afq_decode.hip: per candidate, with per-cell sums)."""
    n = len(w)
    C = candidates(w, bc)
    cs = set(C)
    if 2 not in cs:
        return False, C
    total = 0
    for c in C:
        na = int(w[c])
        if na > n or c + HW + na > n:
            return False, C
        s = c + HW + na
        if s != n and not (s + HW <= n and w[s + 1] == bc):
            return False, C
        total += HW + na
    return len(C) == nrec and total == n - 2, C


def sequential_parse(w, nrec):
    """The reference's walk (one record after the other from dword 2): the record starts, or None when the chunk is malformed."""
    n, i, starts = len(w), 2, []
    while i < n:
        if i + HW > n:
            return None
        starts.append(i)
        i += HW + int(w[i])
    return starts if i == n and len(starts) == nrec else None


def make_chunk(rng, bc, n_records, p_spell):
    """A well-formed chunk whose UMIs / refs / alignment counts spell the barcode with probability p_spell."""
    w = [0, n_records]   # (chunk header: nbytes is filled by nobody here, nrec)
    for _ in range(n_records):
        na = int(rng.integers(0, 7))
        if rng.random() < p_spell and bc < 7:
            na = bc   # an alignment count equal to the barcode value
        w += [na, bc, bc if rng.random() < p_spell else int(rng.integers(0, 50))]
        w += [bc if rng.random() < p_spell else int(rng.integers(0, 50)) for _ in range(na)]
    return np.asarray(w, dtype=np.int64)


def test_proof_accepts_only_the_sequential_parse():
    rng = np.random.default_rng(2024)
    accepted = rejected_well_formed = attacked = 0
    for trial in range(30000):
        bc = int(rng.integers(0, 9))   # small values: na fields and refs collide with it all the time
        nrec = int(rng.integers(1, 9))
        w = make_chunk(rng, bc, nrec, float(rng.choice([0.0, 0.1, 0.4, 0.8])))
        claimed = nrec
        kind = int(rng.integers(0, 5))
        if kind == 1 and len(w) > 6:      # overwrite a few words
            for _ in range(int(rng.integers(1, 4))):
                w[int(rng.integers(2, len(w)))] = int(rng.choice([bc, 0, 1, 2, 3, int(rng.integers(0, 50))]))
            attacked += 1
        elif kind == 2:                   # cut the chunk short / pad it
            cut = int(rng.integers(-3, 4))
            w = w[:len(w) - cut] if cut > 0 else np.concatenate((w, rng.integers(0, 9, -cut)))
            attacked += 1
        elif kind == 3:                   # the header lies about the record count
            claimed = max(1, nrec + int(rng.choice([-1, 1])))
            attacked += 1
        ok, C = proof_holds(w, bc, claimed)
        seq = sequential_parse(w, claimed)
        if ok:
            accepted += 1
            assert seq is not None and C == seq, (trial, bc, w.tolist(), C, seq)
            # ... and every record the walk finds carries the barcode: the property the candidates were found by
            assert all(w[s + 1] == bc for s in seq)
        elif seq is not None and all(w[s + 1] == bc for s in seq):
            # a well-formed chunk of one barcode that the proof turns down: allowed (it goes to the sequential kernel) - it
            # happens only when some other word spells the barcode in a place that makes a false candidate
            rejected_well_formed += 1
            assert set(seq) < set(C), (trial, C, seq)
    assert accepted > 2000 and attacked > 10000 and rejected_well_formed > 100   # (the generator reaches all three regimes: 3 353 / 17 757 / 13 270)


def test_clean_chunks_always_pass():
    """No word but the barcode fields equals the barcode: the proof must hold (nothing takes the slow kernel without a reason)."""
    rng = np.random.default_rng(7)
    for _ in range(3000):
        nrec = int(rng.integers(1, 12))
        w = make_chunk(rng, 1000003, nrec, 0.0)
        ok, C = proof_holds(w, 1000003, nrec)
        assert ok and C == sequential_parse(w, nrec)
