// annotate.cpp -- `conga-annotate`: builds the annotation container `conga --sonic` takes from the inputs a user
// already has: the `--ref` FASTA (chromosome names, lengths, GC% per window) and, optionally, a BED of satellite
// repeats (RepeatMasker / UCSC rmsk style) for the `--rp` path.
//
// Why it exists: the reference loads a `.sonic` file through calkan/sonic (svdepth.c:47, README.md:53-61), an
// un-vendored submodule whose file format is not available here.  What CONGA takes from
// that annotation is exactly three things -- chromosome names and lengths (bam_data.c:269-293), GC% of a 100-base
// window (read_distribution.c:70, likelihood.c:117) and "is this read in a satellite" (bam_data.c:96-97,207) -- and
// all three derive from the FASTA plus a repeat BED.
//
// GC rule: a window is gc_window consecutive bases starting at a multiple of gc_window (the last one is shorter);
// its value is round-half-up of 100 * (#G + #C, either case) / (bases in the window).  Every other letter, N
// included, counts as not-GC, so an assembly gap is GC 0 -- the bin whose expected depth the reference forces to 0
// (read_distribution.c:79).
#include <getopt.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

namespace {

struct chrom {
	std::string name;
	int64_t length = 0;
	std::vector<uint8_t> gc;
	std::vector<std::pair<int32_t, int32_t>> sat; // [start, end), merged, sorted
};

void usage(FILE *f)
{
	fprintf(f, "\nconga-annotate: FASTA (+ satellite BED) -> annotation container for `conga --sonic`\n\n"
			"\t--ref [FASTA]          : reference genome, the one given to conga --ref (required)\n"
			"\t--out [file]           : container to write (required)\n"
			"\t--satellites [BED]     : chr, start, end[, ...] rows of satellite repeats (half-open, 0-based)\n"
			"\t--match [text]         : keep only BED rows with a column (4th or later) containing text, any case\n"
			"\t--gc-window [INT]      : window size in bases (default 100, CONGA's own constant)\n"
			"\t--help\n\n");
}

bool contains_nocase(const std::string &hay, const std::string &needle)
{
	if (needle.empty())
		return true;
	auto it = std::search(hay.begin(), hay.end(), needle.begin(), needle.end(),
			[](char a, char b) { return std::tolower((unsigned char) a) == std::tolower((unsigned char) b); });
	return it != hay.end();
}

// One pass over the FASTA with a large buffer; sequence lines may have any width, '\r' is ignored.
bool scan_fasta(const std::string &path, int gc_window, std::vector<chrom> *out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f)
		return false;
	std::vector<char> buf(1 << 22);
	bool in_header = false, line_start = true;
	std::string header;
	chrom *cur = nullptr;
	int64_t in_win = 0, gc_in_win = 0;
	auto close_window = [&] {
		if (cur && in_win > 0)
			cur->gc.push_back((uint8_t) ((200 * gc_in_win + in_win) / (2 * in_win)));
		in_win = gc_in_win = 0;
	};
	// G, C, g, c -> 1
	uint8_t is_gc[256];
	memset(is_gc, 0, sizeof is_gc);
	is_gc[(unsigned char) 'G'] = is_gc[(unsigned char) 'C'] = is_gc[(unsigned char) 'g'] = is_gc[(unsigned char) 'c'] = 1;
	size_t n;
	while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) {
		for (size_t i = 0; i < n; i++) {
			const char c = buf[i];
			if (in_header) {
				if (c == '\n') {
					in_header = false;
					line_start = true;
					close_window();
					out->emplace_back();
					cur = &out->back();
					const size_t ws = header.find_first_of(" \t\r");
					cur->name = header.substr(0, ws);
				} else {
					header.push_back(c);
				}
				continue;
			}
			if (c == '\n') {
				line_start = true;
				continue;
			}
			if (line_start && c == '>') {
				in_header = true;
				header.clear();
				continue;
			}
			line_start = false;
			if (c == '\r' || !cur)
				continue;
			gc_in_win += is_gc[(unsigned char) c];
			cur->length++;
			if (++in_win == gc_window)
				close_window();
		}
	}
	close_window();
	fclose(f);
	return true;
}

bool load_satellites(const std::string &path, const std::string &match, std::vector<chrom> *chroms, long *kept, long *skipped)
{
	FILE *f = fopen(path.c_str(), "r");
	if (!f)
		return false;
	std::map<std::string, chrom *> by_name;
	for (chrom &c : *chroms)
		by_name[c.name] = &c;
	char *line = nullptr;
	size_t cap = 0;
	while (getline(&line, &cap, f) > 0) {
		if (line[0] == '#' || strncmp(line, "track", 5) == 0 || strncmp(line, "browser", 7) == 0)
			continue;
		std::vector<std::string> col;
		for (char *tok = strtok(line, " \t\r\n"); tok; tok = strtok(nullptr, " \t\r\n"))
			col.emplace_back(tok);
		if (col.size() < 3)
			continue;
		bool keep = match.empty();
		for (size_t k = 3; k < col.size() && !keep; k++)
			keep = contains_nocase(col[k], match);
		auto it = by_name.find(col[0]);
		if (!keep || it == by_name.end()) {
			(*skipped)++;
			continue;
		}
		int64_t s = strtoll(col[1].c_str(), nullptr, 10), e = strtoll(col[2].c_str(), nullptr, 10);
		s = std::max<int64_t>(s, 0);
		e = std::min<int64_t>(e, it->second->length);
		if (e <= s) {
			(*skipped)++;
			continue;
		}
		it->second->sat.emplace_back((int32_t) s, (int32_t) e);
		(*kept)++;
	}
	free(line);
	fclose(f);
	for (chrom &c : *chroms) {
		std::sort(c.sat.begin(), c.sat.end());
		std::vector<std::pair<int32_t, int32_t>> merged;
		for (const auto &iv : c.sat) {
			if (!merged.empty() && iv.first <= merged.back().second)
				merged.back().second = std::max(merged.back().second, iv.second);
			else
				merged.push_back(iv);
		}
		c.sat.swap(merged);
	}
	return true;
}

bool write_container(const std::string &path, int gc_window, const std::vector<chrom> &chroms)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f)
		return false;
	bool ok = fwrite("CONGAAN1", 1, 8, f) == 8;
	const int32_t hdr[2] = {gc_window, (int32_t) chroms.size()};
	ok = ok && fwrite(hdr, 4, 2, f) == 2;
	for (const chrom &c : chroms) {
		const uint16_t ln = (uint16_t) c.name.size();
		const int64_t v[3] = {c.length, (int64_t) c.gc.size(), (int64_t) c.sat.size()};
		ok = ok && fwrite(&ln, 2, 1, f) == 1 && fwrite(c.name.data(), 1, ln, f) == ln && fwrite(v, 8, 3, f) == 3;
	}
	for (const chrom &c : chroms) {
		std::vector<int32_t> s(c.sat.size()), e(c.sat.size());
		for (size_t i = 0; i < c.sat.size(); i++) {
			s[i] = c.sat[i].first;
			e[i] = c.sat[i].second;
		}
		ok = ok && fwrite(c.gc.data(), 1, c.gc.size(), f) == c.gc.size();
		if (!s.empty()) // (fwrite's pointer argument must not be null, even for zero elements)
			ok = ok && fwrite(s.data(), 4, s.size(), f) == s.size() && fwrite(e.data(), 4, e.size(), f) == e.size();
	}
	return fclose(f) == 0 && ok;
}

} // namespace

int main(int argc, char **argv)
{
	static struct option long_options[] = {{"ref", required_argument, 0, 'f'}, {"out", required_argument, 0, 'o'},
		{"satellites", required_argument, 0, 's'}, {"match", required_argument, 0, 'm'}, {"gc-window", required_argument, 0, 'w'},
		{"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
	std::string ref, out, sat, match;
	int gc_window = 100, o, index = 0;
	while ((o = getopt_long(argc, argv, "f:o:s:m:w:h", long_options, &index)) != -1) {
		switch (o) {
		case 'f': ref = optarg; break;
		case 'o': out = optarg; break;
		case 's': sat = optarg; break;
		case 'm': match = optarg; break;
		case 'w': gc_window = atoi(optarg); break;
		case 'h': usage(stdout); return 0;
		default: usage(stderr); return 3;
		}
	}
	if (ref.empty() || out.empty()) {
		fprintf(stderr, "[CONGA CMDLINE ERROR] conga-annotate needs --ref and --out.\n");
		usage(stderr);
		return 3;
	}
	if (gc_window <= 0 || gc_window > 1024) {
		fprintf(stderr, "[CONGA CMDLINE ERROR] --gc-window must be in 1..1024.\n");
		return 3;
	}
	std::vector<chrom> chroms;
	if (!scan_fasta(ref, gc_window, &chroms)) {
		fprintf(stderr, "\n[CONGA INPUT ERROR] Unable to open file %s in read mode.\n", ref.c_str());
		return 1;
	}
	chroms.erase(std::remove_if(chroms.begin(), chroms.end(), [](const chrom &c) {
		if (c.length == 0)
			fprintf(stderr, "[CONGA] sequence %s is empty: left out\n", c.name.c_str());
		return c.length == 0;
	}), chroms.end());
	if (chroms.empty()) {
		fprintf(stderr, "\n[CONGA INPUT ERROR] %s holds no sequence.\n", ref.c_str());
		return 1;
	}
	if (chroms.size() > 1024) {
		fprintf(stderr, "\n[CONGA INPUT ERROR] %s holds %zu sequences; the container takes at most 1024.\n", ref.c_str(), chroms.size());
		return 1;
	}
	for (const chrom &c : chroms)
		if (c.length > INT32_MAX) {
			fprintf(stderr, "\n[CONGA INPUT ERROR] sequence %s is longer than 2^31-1 bases.\n", c.name.c_str());
			return 1;
		}
	long kept = 0, skipped = 0;
	if (!sat.empty() && !load_satellites(sat, match, &chroms, &kept, &skipped)) {
		fprintf(stderr, "\n[CONGA INPUT ERROR] Unable to open file %s in read mode.\n", sat.c_str());
		return 1;
	}
	if (!write_container(out, gc_window, chroms)) {
		fprintf(stderr, "\n[CONGA INPUT ERROR] Unable to open file %s in write mode.\n", out.c_str());
		return 1;
	}
	printf("#chromosome\tlength\twindows\tmean_gc\tsatellite_intervals\n");
	for (const chrom &c : chroms) {
		double sum = 0;
		for (uint8_t g : c.gc)
			sum += g;
		printf("%s\t%lld\t%zu\t%.2f\t%zu\n", c.name.c_str(), (long long) c.length, c.gc.size(), c.gc.empty() ? 0.0 : sum / c.gc.size(),
				c.sat.size());
	}
	if (!sat.empty())
		printf("#satellite rows kept %ld, skipped %ld (filtered out, unknown chromosome or empty after clipping)\n", kept, skipped);
	return 0;
}
