// serial_f32.h -- exact fast-forward of the reference's serial float32 accumulator.
//
// calculate_likelihood_CNV adds expected_read_depth[gc] to a float once per BASE, left to right
// (/root/reference/likelihood.c:111,115-119).  All bases of one GC window add the same value c, so
// the loop body is "s = fl32(s + c)" repeated k times.  Within one binade of s every such add
// moves s by the same whole number of ulps (round-to-nearest-even of c/ulp(s)), so k adds collapse
// to one integer multiply-add; only binade crossings and the first step of an exact tie need a
// real add.  The result is bit-identical to k hardware adds (checked exhaustively-by-property in
// tests/test_serial_f32.py against the literal loop).
//
// Operands must be finite.  The fast path needs s and c of equal sign (expected_read_depth is a
// non-negative table unless a depth counter wrapped, read_distribution.c:75-83); sub-normal or
// mixed-sign operands fall back to real adds.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CONGA_HD __host__ __device__ __forceinline__
#else
#define CONGA_HD static inline
#endif

CONGA_HD uint32_t conga_f32_bits(float x)
{
	uint32_t u;
	__builtin_memcpy(&u, &x, sizeof u);
	return u;
}

CONGA_HD float conga_bits_f32(uint32_t u)
{
	float x;
	__builtin_memcpy(&x, &u, sizeof x);
	return x;
}

// Per-(binade, addend) step: how many ulps of s one add of c advances s while s stays in the
// binade with biased exponent `es` and keeps an even mantissa when the add is an exact tie.
//   q      = floor(c / ulp), lim = 2^24 - 1 - q (largest mantissa from which an add stays regular)
//   returns delta, or 0xFFFFFFFF when the pair is outside the regular regime (caller does real adds)
struct conga_step {
	uint32_t delta;
	uint32_t lim;
	uint32_t tie; // 1: exact tie -- delta is valid only from an even mantissa
};

CONGA_HD conga_step conga_step_for(uint32_t es, uint32_t bits_c)
{
	conga_step st;
	const uint32_t ec = bits_c >> 23;
	st.delta = 0xFFFFFFFFu;
	st.lim = 0;
	st.tie = 0;
	if (bits_c == 0) { // adding +0 never changes s
		st.delta = 0;
		st.lim = 0xFFFFFFu;
		return st;
	}
	if (es == 0 || ec == 0 || es <= ec || es == 255)
		return st;
	const uint32_t d = es - ec;
	if (d >= 25) { // c < ulp/2: s is stuck
		st.delta = 0;
		st.lim = 0xFFFFFFu;
		return st;
	}
	const uint32_t mc = (bits_c & 0x7FFFFFu) | 0x800000u;
	const uint32_t q = mc >> d;
	const uint32_t r = mc & ((1u << d) - 1u);
	const uint32_t half = 1u << (d - 1);
	st.lim = 0xFFFFFFu - q;
	if (r < half)
		st.delta = q;
	else if (r > half)
		st.delta = q + 1u;
	else {
		st.delta = q + (q & 1u);
		st.tie = 1;
	}
	return st;
}

// s after `k` repetitions of s = fl32(s + c), for s >= 0 and c >= 0.
CONGA_HD float conga_repeat_add_nonneg_f32(float s, float c, uint32_t k)
{
	const uint32_t bc = conga_f32_bits(c);
	while (k) {
		const uint32_t bs = conga_f32_bits(s);
		const uint32_t es = bs >> 23;
		const conga_step st = conga_step_for(es, bc);
		uint32_t ms = (bs & 0x7FFFFFu) | 0x800000u;
		if (st.delta == 0xFFFFFFFFu || ms > st.lim || (st.tie && (ms & 1u))) {
			s = s + c; // binade crossing, sub-normal operand, or first step of a tie from an odd mantissa
			--k;
			continue;
		}
		if (st.delta == 0)
			return s;
		// all k adds stay regular iff the last one starts at or below lim; only a run that reaches the
		// binade top needs the (slow, rare) integer divide
		uint32_t n = k;
		if ((uint64_t) (k - 1u) * st.delta > (uint64_t) (st.lim - ms))
			n = (st.lim - ms) / st.delta + 1u;
		ms += n * st.delta; // <= 2^24
		k -= n;
		s = (ms == 0x1000000u) ? conga_bits_f32((es + 1u) << 23) : conga_bits_f32((es << 23) | (ms & 0x7FFFFFu));
	}
	return s;
}

// Any finite operands.  expected_read_depth can only go negative when a `short` depth counter has
// wrapped (more than 32767 read starts on one base); round-to-nearest-even is symmetric, so equal
// signs reuse the non-negative routine and mixed signs fall back to literal adds.
CONGA_HD float conga_repeat_add_f32(float s, float c, uint32_t k)
{
	while (k) {
		const uint32_t bs = conga_f32_bits(s), bc = conga_f32_bits(c);
		const bool s_neg = (bs >> 31) != 0, c_neg = (bc >> 31) != 0;
		if (!s_neg && !c_neg)
			return conga_repeat_add_nonneg_f32(s, c, k);
		if (s_neg && c_neg)
			return -conga_repeat_add_nonneg_f32(-s, -c, k);
		if ((bc << 1) == 0)
			return s + c; // +-0 addend: one add settles the sign of a zero accumulator, later ones change nothing
		s = s + c;    // opposite signs (or a signed zero accumulator): one literal add, then look again
		--k;
	}
	return s;
}
