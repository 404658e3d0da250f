"""A second, independent restatement of the reference's split-read rules, in plain Python (strings, dicts, lists) --
written from the reference's text, not from oracle/conga_oracle_sr.c, so that rows a15-a18 of SURVEY.md section 8 are
not single-sourced.  Small cases only (it is a few thousand reads per second).

  10-mer index       split_read.c:357-466  every start whose ten letters are all ACGT, in increasing order; buckets that
                                           are empty or hold >= MAX_SR_HIT (50 000) starts are dropped
  the gate           bam_data.c:205-207    qual > mq, l_qseq > min_read_length, is_proper (common.c:317-323), no
                                           satellite in [pos, pos + 20)
  find_split_reads   split_read.c:206-354  pos == 0 skipped; element 1 = second half at pos, element 2 = first half at
                                           pos + l/2; the float quality sum is NOT reset between the two
  half-read mapping  split_read.c:75-204   forward bucket scan, then (if < MAX_MAPPING hits) the reverse complement's, hits
                                           within SR_LOOKAHEAD of the anchor and Hamming <= (int)(0.05 * len); the reverse
                                           scan stops once more than MAX_MAPPING are held; 0 < hits < MAX_MAPPING kept,
                                           mapq = 60 / hits
  pairing            bam_data.c:29-154     satellite / mapq / bounds filters, forward-forward only, DEL / DUP by which
                                           half maps where, 50-base trim (SOFTCLIP_WRONGMAP_WINDOW)
  support            likelihood.c:41-94    count_ReadPairs windows (WRONGMAP_WINDOW 100, WRONGMAP_WINDOW_DEL 5000)
"""
import numpy as np

K = 10
MAX_SR_HIT, MAX_MAPPING, SR_LOOKAHEAD = 50_000, 100, 100_000
SOFTCLIP, WRONG, WRONG_DEL = 50, 100, 5000
LETTER = {1: "A", 2: "C", 4: "G", 8: "T", 15: "N"}
COMP = {"A": "T", "T": "A", "G": "C", "C": "G", "N": "N"}


def f32(x):
    return np.float32(x)


def build_index(ref):
    """ref: str, upper case.  -> {10-mer: [starts]} without the dropped buckets."""
    index = {}
    ok = set("ACGT")
    bad_until = -1   # index of the last non-ACGT letter seen so far
    for j, ch in enumerate(ref):
        if ch not in ok:
            bad_until = j
        i = j - K + 1
        if i >= 0 and bad_until < i:
            index.setdefault(ref[i:j + 1], []).append(i)
    return {k: v for k, v in index.items() if len(v) < MAX_SR_HIT}


def satellite(sats, a, b):
    return 1 if any(s < b and e > a for s, e in sats) else 0


def map_half(ref, index, s, anchor):
    n = len(s)
    if n < K:
        return []
    dist_max = int(0.05 * n)
    hits = []

    def scan(text, orient, stop_past_max):
        for p in index.get(text[:K], ()) if set(text[:K]) <= set("ACGT") else ():
            if abs(p - anchor) < SR_LOOKAHEAD:
                window = ref[p:p + n]
                d = sum(1 for x, y in zip(window, text) if x != y) + (n - len(window))
                if d <= dist_max:
                    hits.append((p, orient))
            if stop_past_max and len(hits) > MAX_MAPPING:
                break

    scan(s, "F", False)
    if len(hits) < MAX_MAPPING:
        scan("".join(COMP[c] for c in reversed(s)), "R", True)
    if not (0 < len(hits) < MAX_MAPPING):
        return []
    q = 60 // len(hits)
    return [(p, o, q) for p, o in hits]


def split_read_rows(ref, sats, reads, mq_threshold=-1, min_read_length=60):
    """reads: [(pos, mapq, flag, codes (4-bit values), quals)] in file order.
    -> (rows [(type, left_end, right_start)], (elements, mappings, del rows, dup rows))"""
    L = len(ref)
    index = build_index(ref)
    rows, n_elem, n_map = [], 0, 0
    for pos, mapq, flag, codes, quals in reads:
        l = len(codes)
        if not (mapq > mq_threshold and l > min_read_length and (flag & (0x100 | 0x800 | 0x400 | 0x200)) == 0):
            continue
        if satellite(sats, pos, pos + 20) or pos == 0:
            continue
        text = "".join(LETTER[int(c)] for c in codes)
        half = l // 2
        avg = f32(0)
        for which, (lo, hi, anchor) in enumerate(((half, l, pos), (0, half, pos + half))):
            for q in quals[lo:hi]:
                avg = f32(avg + f32(q))
            avg = f32(avg / f32(hi - lo))
            if int(np.floor(avg)) < mq_threshold:
                break
            n_elem += 1
            maps = map_half(ref, index, text[lo:hi], anchor)
            n_map += len(maps)
            for p, orient, q in maps:
                if satellite(sats, anchor, anchor + 1) + satellite(sats, p, p + 1) != 0:
                    continue
                if not (q > mq_threshold and anchor > 0 and p > 0 and anchor < L and p < L) or p == anchor:
                    continue
                len_split, len_read = l // 2, l - l // 2
                if anchor < p:
                    one_end, two_start = anchor + len_read, p
                else:
                    one_end, two_start = p + len_split, anchor
                if one_end >= two_start or orient != "F":
                    continue
                second = which == 1   # the "_read2" element
                kind = "D" if (anchor < p) != second else "E"
                rows.append((kind, one_end - SOFTCLIP, two_start + SOFTCLIP))
    n_del = sum(1 for r in rows if r[0] == "D")
    return rows, (n_elem, n_map, n_del, len(rows) - n_del)


def count_read_pairs(rows, dels, dups):
    """-> (border_rp per deletion, rp per duplication)"""
    border = [0] * len(dels)
    rp = [0] * len(dups)
    for kind, left_end, right_start in rows:
        if kind == "E":
            for i, (s, e) in enumerate(dups):
                if s - WRONG_DEL <= left_end <= e + WRONG_DEL and s - WRONG_DEL <= right_start <= e + WRONG_DEL:
                    rp[i] += 1
        else:
            for i, (s, e) in enumerate(dels):
                if s - WRONG_DEL <= left_end <= s + WRONG and e - WRONG <= right_start <= e + WRONG_DEL:
                    border[i] += 1
    return border, rp
