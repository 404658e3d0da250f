// split_geom.h -- index arithmetic of the split-read comparison's wide loads (split_map.hip.h), host- and device-callable.
//
// half_distance_rev compares a half read's reverse complement with the reference window [c, c + n) 56 bases per pair of 16-byte
// loads, from the window's END backwards: step k0 (0, 56, 112, ...) looks at the window's bases [c + n - k0 - m, c + n - k0),
// m = min(56, n - k0), and LOADS the 56 bases that end there -- from base sr_rev_step_base(c, n, k0) on.  The last step's m is
// (n - 1) % 56 + 1, so its load begins up to 55 bases in front of the window: in front of the reference text's first base when c
// is small.  The caller may take the wide form only when sr_rev_wide_ok(c, n) says every load begins inside the text
// (almostPerfect_match_seq_ref's reverse-complement branch, split_read.c:158-203: the reference compares base by base).
//
// Round 3's guard was `c + n >= 64` -- right for halves of up to 56 bases, eight bases (a dword) to spare; a half of more than 56
// bases whose reverse complement maps within 56 of chromosome 1's first base loaded two dwords in front of the buffer
// (profiles/r03l_split_map_fault_seed82_case11.log).  tests/test_split_geom.py walks every (c, n) on the host.
#pragma once

#if defined(__HIPCC__)
#define CONGA_SR_HD __host__ __device__ __forceinline__
#else
#define CONGA_SR_HD inline
#endif

// first base of the 56 that step k0 loads (may be negative: then the wide form must not be used)
CONGA_SR_HD int sr_rev_step_base(int c, int n, int k0)
{
	return c + n - k0 - 56;
}

// every load of half_distance_rev(c, n) begins at base 8 or behind it (a dword to spare in front, as the old test had)
CONGA_SR_HD bool sr_rev_wide_ok(int c, int n)
{
	return c + (n - 1) % 56 + 1 >= 56 + 8;
}
