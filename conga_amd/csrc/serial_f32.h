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

// The general routine is the rare path of every kernel that uses it: keeping it out of line keeps the chain
// kernel's loops small enough for the instruction cache.
#if defined(__HIPCC__)
#define CONGA_HD_RARE __host__ __device__ inline __attribute__((noinline))
#else
#define CONGA_HD_RARE static inline
#endif

// product of two operands below 2^24 whose result fits 32 bits: the full-rate 24-bit multiplier on the device
#if defined(__HIP_DEVICE_COMPILE__)
#define CONGA_MUL24(a, b) __umul24((a), (b))
#else
#define CONGA_MUL24(a, b) ((uint32_t) (a) * (uint32_t) (b))
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
CONGA_HD_RARE float conga_repeat_add_f32(float s, float c, uint32_t k)
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

// ---- the lean forms the chain kernels use ---------------------------------------------------------------------
// Same arithmetic as conga_step_for, split so that the part that depends only on the addend is done once per GC
// window and the part that depends on the accumulator's binade is a dozen select-style operations without a
// branch (the chain kernels evaluate it for every window of every pass).
struct conga_addend {
	uint32_t ec;   // biased exponent
	uint32_t mc;   // 24-bit significand
	uint32_t zero; // addend is +0
	uint32_t bad;  // negative, sub-normal, inf / nan: only the general routine handles it
};

CONGA_HD conga_addend conga_addend_of(uint32_t bits_c)
{
	conga_addend c;
	c.ec = (bits_c >> 23) & 0xFFu;
	c.mc = (bits_c & 0x7FFFFFu) | 0x800000u;
	c.zero = (bits_c == 0u) ? 1u : 0u;
	c.bad = ((bits_c >> 31) != 0u || (c.ec == 0u && bits_c != 0u) || c.ec == 255u) ? 1u : 0u;
	return c;
}

struct conga_lean_step {
	uint32_t delta;
	uint32_t lim;
	uint32_t tie; // exact tie: delta holds only from an even mantissa
	uint32_t ok;  // 0: outside the regular regime (or a step above 2^21 ulps, which would not fit the 24-bit multiplies)
};

CONGA_HD conga_lean_step conga_step_lean(uint32_t es, const conga_addend &c)
{
	conga_lean_step st;
	const int d = (int) es - (int) c.ec;
	const bool stuck = c.zero != 0u || d >= 25; // adding +0, or an addend below half an ulp
	const bool reg = d >= 1 && d <= 24;
	const uint32_t dd = (uint32_t) ((d < 1) ? 1 : (d > 24 ? 24 : d));
	const uint32_t q = c.mc >> dd;
	const uint32_t r = c.mc & ((1u << dd) - 1u);
	const uint32_t half = 1u << (dd - 1u);
	const uint32_t tie = (r == half) ? 1u : 0u;
	st.delta = stuck ? 0u : q + ((r > half) ? 1u : 0u) + (tie & q);
	st.lim = stuck ? 0xFFFFFFu : 0xFFFFFFu - q;
	st.tie = stuck ? 0u : tie;
	st.ok = (c.bad == 0u && es >= 1u && es <= 254u && (stuck || reg) && st.delta <= (1u << 21)) ? 1u : 0u;
	return st;
}

// bits of the float with biased exponent es and 24-bit significand ms in [2^23, 2^24] (2^24 = the next binade's 1.0)
CONGA_HD float conga_compose_f32(uint32_t es, uint32_t ms)
{
	return conga_bits_f32((es << 23) + (ms - 0x800000u));
}

// The same result as conga_repeat_add_f32 for k <= 1024 adds (one GC window) of a
// non-negative normal addend.  One loop iteration per binade the accumulator passes through: the regular adds in
// front of the binade top are one integer step -- their number is a quotient below 1024, so a float estimate is at
// most one off and two 24-bit multiplies settle it, no integer divide -- then one real add carries the accumulator
// across.  Wherever the integer step does not apply (accumulator still within 8x of the addend, which includes the
// start from zero; a tie from an odd mantissa) the iteration is one literal add, which is always right.  Only
// negative, sub-normal, infinite operands or k > 1024 go to the general routine.
CONGA_HD_RARE float conga_window_add_loop_f32(float s, float c, uint32_t k)
{
	const uint32_t bc = conga_f32_bits(c);
	const conga_addend ca = conga_addend_of(bc);
	if (ca.bad != 0u || k > 1024u || (conga_f32_bits(s) >> 31) != 0u)
		return conga_repeat_add_f32(s, c, k);
	while (k != 0u) {
		const uint32_t bs = conga_f32_bits(s);
		const uint32_t es = bs >> 23;
		if (es >= 255u)
			return conga_repeat_add_f32(s, c, k); // inf / nan
		const conga_lean_step st = conga_step_lean(es, ca);
		uint32_t ms = (bs & 0x7FFFFFu) | 0x800000u;
		if (st.ok == 0u || (st.tie != 0u && (ms & 1u) != 0u)) {
			s = s + c; // outside the regular regime: the literal operation
			--k;
			continue;
		}
		if (st.delta == 0u)
			return s; // +0, or an addend below half an ulp: stuck
		const bool room_ok = ms <= st.lim;
		const uint32_t room = room_ok ? st.lim - ms : 0u; // < 2^24
		if (room_ok && CONGA_MUL24(k - 1u, st.delta) <= room)
			return conga_compose_f32(es, ms + CONGA_MUL24(k, st.delta)); // the rest of the window inside this binade
		uint32_t n1 = 0; // regular adds in front of the crossing: the last one starts at or below lim
		if (room_ok) {
			uint32_t q = (uint32_t) ((float) room / (float) st.delta); // true quotient < k - 1 <= 1023
			if (CONGA_MUL24(q, st.delta) > room)
				q--;
			else if (CONGA_MUL24(q + 1u, st.delta) <= room)
				q++;
			n1 = q + 1u;
			ms += CONGA_MUL24(n1, st.delta); // lim < ms <= 2^24
		}
		s = conga_compose_f32(es, ms) + c; // the add that reaches or crosses the binade top: real rounding
		k -= n1 + 1u;
	}
	return s;
}

// What the chain kernels call for every window.  Straight-line, select-style code for the two cases that make up
// all but the first windows of a chain -- the whole window inside the accumulator's binade, or exactly one crossing of
// its top -- because in the lane-per-interval class every lane of the wave sits in this routine once per trip and a
// branch taken by one lane is paid by all 64 (branches and their mask bookkeeping, not arithmetic, were most of a
// trip).  Both candidates are always computed; everything else (start from zero, ties from odd mantissas, two
// crossings in one window, odd operands) goes to the loop above through one rarely taken branch.
struct conga_window_head { // stage 1 of conga_window_add_f32: the accumulator's own binade
	conga_addend ca;
	conga_lean_step st;
	uint32_t es, ms, room, dl, kk;
	bool ok1, room_ok;
	bool fit;      // ok1 and the whole window stays inside the binade: res_fit is the answer
	float res_fit;
};

CONGA_HD conga_window_head conga_window_stage1(float s, float c, uint32_t k)
{
	conga_window_head h;
	const uint32_t bs = conga_f32_bits(s);
	h.ca = conga_addend_of(conga_f32_bits(c));
	h.es = bs >> 23; // a negative accumulator shows up as es >= 256: not ok
	h.st = conga_step_lean(h.es, h.ca);
	h.ms = (bs & 0x7FFFFFu) | 0x800000u;
	h.ok1 = h.st.ok != 0u && k <= 1024u && !(h.st.tie != 0u && (h.ms & 1u) != 0u);
	h.room_ok = h.ms <= h.st.lim;
	h.room = h.room_ok ? h.st.lim - h.ms : 0u; // < 2^24
	h.dl = h.st.delta & 0x3FFFFFu;              // (<= 2^21 whenever ok)
	h.kk = k & 0x7FFu;                          // (<= 1024 whenever ok)
	h.fit = h.ok1 && h.room_ok && CONGA_MUL24((h.kk - 1u) & 0x7FFu, h.dl) <= h.room;
	h.res_fit = conga_compose_f32(h.es & 0xFFu, h.ms + CONGA_MUL24(h.kk, h.dl));
	return h;
}

// Everything behind stage 1 (k != 0): the one-crossing candidate, then the loop.
CONGA_HD float conga_window_finish(const conga_window_head &h, float s, float c, uint32_t k)
{
	const uint32_t es = h.es, ms = h.ms, room = h.room, dl = h.dl, kk = h.kk;
	// ---- one crossing: n1 regular adds (the last one starts at or below lim), one real add, the rest in the next binade
	uint32_t q = (uint32_t) ((float) room / (float) (dl | (dl == 0u ? 1u : 0u))); // true quotient < k - 1 <= 1023 when it matters
	q = (q > 2047u) ? 2047u : q;
	const bool q_high = CONGA_MUL24(q, dl) > room;
	const bool q_low = !q_high && CONGA_MUL24(q + 1u, dl) <= room;
	q = q_high ? q - 1u : (q_low ? q + 1u : q);
	const uint32_t n1 = h.room_ok ? q + 1u : 0u;
	const uint32_t ms1 = ms + CONGA_MUL24(n1 & 0x7FFu, dl); // lim < ms1 <= 2^24 in the case that is used
	const float s2 = conga_compose_f32(es & 0xFFu, ms1) + c; // the add that reaches or crosses the binade top: real rounding
	const uint32_t r = kk - n1 - 1u;                          // adds left (wraps when the window fits: unused then)
	// ---- stage 2: the next binade
	const uint32_t bs2 = conga_f32_bits(s2);
	const uint32_t es2 = bs2 >> 23;
	const conga_lean_step st2 = conga_step_lean(es2, h.ca);
	const uint32_t ms2 = (bs2 & 0x7FFFFFu) | 0x800000u;
	const uint32_t dl2 = st2.delta & 0x3FFFFFu;
	const bool ok2 = st2.ok != 0u && !(st2.tie != 0u && (ms2 & 1u) != 0u) && ms2 <= st2.lim
			&& CONGA_MUL24((r - 1u) & 0x7FFu, dl2) <= st2.lim - ms2;
	const float res_cross = (r == 0u) ? s2 : conga_compose_f32(es2 & 0xFFu, ms2 + CONGA_MUL24(r & 0x7FFu, dl2));
	const bool cross_ok = h.st.delta != 0u && n1 < kk && (r == 0u || ok2);
	if (h.fit || (h.ok1 && cross_ok))
		return h.fit ? h.res_fit : res_cross;
	return conga_window_add_loop_f32(s, c, k);
}

CONGA_HD float conga_window_add_f32(float s, float c, uint32_t k)
{
	if (k == 0u)
		return s;
	return conga_window_finish(conga_window_stage1(s, c, k), s, c, k);
}
