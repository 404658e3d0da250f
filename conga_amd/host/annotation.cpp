#include "annotation.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

namespace conga_host {

namespace {
bool read_exact(FILE *f, void *dst, size_t n)
{
	return n == 0 || fread(dst, 1, n, f) == n;
}
} // namespace

sonic *sonic_load(const std::string &path, std::string *err)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f) {
		*err = "cannot open annotation file " + path;
		return nullptr;
	}
	std::unique_ptr<sonic> s(new sonic());
	char magic[8];
	int32_t hdr[2];
	bool ok = read_exact(f, magic, 8) && memcmp(magic, "CONGAAN1", 8) == 0 && read_exact(f, hdr, 8);
	if (!ok || hdr[0] <= 0 || hdr[0] > 1024 || hdr[1] < 0) {
		*err = path + " is not a CONGAAN1 annotation container (real .sonic files are not readable here: "
				"calkan/sonic is absent; build one with conga_amd.formats.write_annotation)";
		fclose(f);
		return nullptr;
	}
	s->gc_step = hdr[0];
	s->number_of_chromosomes = hdr[1];
	std::vector<int64_t> n_win(hdr[1]), n_sat(hdr[1]);
	for (int c = 0; ok && c < hdr[1]; c++) {
		uint16_t ln = 0;
		int64_t v[3];
		ok = read_exact(f, &ln, 2);
		std::string name(ln, '\0');
		ok = ok && read_exact(f, &name[0], ln) && read_exact(f, v, 24);
		if (ok && (v[0] <= 0 || v[1] != (v[0] + s->gc_step - 1) / s->gc_step || v[2] < 0))
			ok = false;
		s->chromosome_names.push_back(name);
		s->chromosome_lengths.push_back(v[0]);
		n_win[c] = v[1];
		n_sat[c] = v[2];
	}
	s->gc.resize(hdr[1]);
	s->sat_start.resize(hdr[1]);
	s->sat_end.resize(hdr[1]);
	for (int c = 0; ok && c < hdr[1]; c++) {
		s->gc[c].resize((size_t) n_win[c]);
		s->sat_start[c].resize((size_t) n_sat[c]);
		s->sat_end[c].resize((size_t) n_sat[c]);
		ok = read_exact(f, s->gc[c].data(), (size_t) n_win[c]) && read_exact(f, s->sat_start[c].data(), (size_t) n_sat[c] * 4)
				&& read_exact(f, s->sat_end[c].data(), (size_t) n_sat[c] * 4);
	}
	fclose(f);
	if (!ok) {
		*err = "truncated or corrupt annotation container " + path;
		return nullptr;
	}
	return s.release();
}

int sonic_refind_chromosome_index(const sonic *s, const std::string &chr)
{
	for (int c = 0; c < s->number_of_chromosomes; c++)
		if (s->chromosome_names[c] == chr)
			return c;
	return -1;
}

// THE assumed sonic rule: the GC% of the gc_step-base window that holds `start`, window index
// clamped to the chromosome's windows.  gc_of_window is the same rule for a caller that already has the window index.
static inline float gc_of_window(const std::vector<uint8_t> &g, int64_t w)
{
	if (w >= (int64_t) g.size())
		w = (int64_t) g.size() - 1;
	if (w < 0)
		w = 0;
	return (float) g[(size_t) w];
}

float sonic_get_gc_content(const sonic *s, int chr_index, int64_t start, int64_t end)
{
	(void) end;
	return gc_of_window(s->gc[chr_index], start / s->gc_step);
}

int sonic_is_satellite(const sonic *s, int chr_index, int64_t start, int64_t end)
{
	// any satellite interval overlapping [start, end)
	const std::vector<int32_t> &ss = s->sat_start[chr_index], &se = s->sat_end[chr_index];
	size_t lo = 0, hi = ss.size();
	while (lo < hi) {
		const size_t mid = (lo + hi) / 2;
		if ((int64_t) se[mid] <= start)
			lo = mid + 1;
		else
			hi = mid;
	}
	return (lo < ss.size() && (int64_t) ss[lo] < end) ? 1 : 0;
}

// (int) round(x) for the non-negative GC percentages of read_distribution.c:70 / likelihood.c:117, without the libm
// call (29 million windows, two lookups each, per genome): truncate, then half away from zero on the exact remainder
static inline int round_gc(float x)
{
	if (!(x >= 0.0f))
		return (int) std::round(x);
	const int t = (int) x;
	return t + ((x - (float) t >= 0.5f) ? 1 : 0);
}

void gc_window_arrays(const sonic *s, int chr_index, std::vector<uint8_t> *gc_hist_w, std::vector<uint8_t> *gc_like_w)
{
	const int64_t L = s->chromosome_lengths[chr_index];
	const int64_t step = s->gc_step;
	const int64_t n_win = (L + step - 1) / step;
	gc_hist_w->resize((size_t) n_win);
	gc_like_w->resize((size_t) n_win);
	const std::vector<uint8_t> &g = s->gc[chr_index];
	if ((int64_t) g.size() == n_win) {
		// The container holds one rounded percentage per window: (int) round((float) byte) is the byte, and both lookups
		// of window w read entry w -- the loop below would copy the track twice, one libm-free rounding per window.
		*gc_hist_w = g;
		*gc_like_w = g;
		return;
	}
	for (int64_t w = 0; w < n_win; w++) {
		// The two lookups of the reference for the bases i = w * step of this window:
		//   read_distribution.c:65-70  sonic_get_gc_content(chr, i, min(i + step, L))   (end clamped to the chromosome)
		//   likelihood.c:117           sonic_get_gc_content(chr, i, i + step)           (end not clamped)
		// Under the assumed rule neither depends on `end`, and start / step is w itself (29 million windows per genome:
		// the two 64-bit divisions per window were 80 ms of a 250 ms run).
		(*gc_hist_w)[(size_t) w] = (uint8_t) round_gc(gc_of_window(g, w));
		(*gc_like_w)[(size_t) w] = (uint8_t) round_gc(gc_of_window(g, w));
	}
	(void) L;
}

} // namespace conga_host
