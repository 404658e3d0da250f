#include "svs.h"
#include "knobs.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace conga_host {

namespace {
const char *kRowDelimiters = " \t\r\n"; // svs.h:7 ROW_DELIMITERS
const int kLineChunk = 512;              // svs.c:11

bool blank(const char *line)
{
	for (; *line; line++)
		if (!isspace((unsigned char) *line))
			return false;
	return true;
}
} // namespace

// The reference's reader, literally: 512-byte fgets chunks, strtok, atoi, atof.  Used when a line is long enough
// for the chunking to matter (>= 511 characters) and as the specification of the fast path below.
static bool load_bed_fgets(const std::string &path, bool with_value, bed_index *out)
{
	FILE *f = fopen(path.c_str(), "r");
	if (!f)
		return false;
	char line[kLineChunk];
	std::string last_chr;
	std::vector<sv_row> *rows = nullptr;
	std::vector<float> *vals = nullptr;
	while (fgets(line, kLineChunk, f) != nullptr) {
		if (blank(line))
			continue;
		char *save = nullptr;
		const char *chr = strtok_r(line, kRowDelimiters, &save);
		const char *ts = strtok_r(nullptr, kRowDelimiters, &save);
		const char *te = strtok_r(nullptr, kRowDelimiters, &save);
		const char *tv = with_value ? strtok_r(nullptr, kRowDelimiters, &save) : nullptr;
		if (!chr || !ts || !te || (with_value && !tv))
			continue; // the reference would crash in atoi(NULL)
		if (!rows || last_chr != chr) {
			last_chr = chr;
			rows = &out->rows[last_chr];
			vals = with_value ? &out->values[last_chr] : nullptr;
		}
		rows->push_back(sv_row{atoi(ts), atoi(te)});
		if (vals)
			vals->push_back((float) atof(tv)); // svs.c:365: float mappability = atof(...)
	}
	fclose(f);
	return true;
}

namespace {

inline bool is_delim(char c)
{
	return c == ' ' || c == '\t' || c == '\r' || c == '\n';
}

// atoi on a token [b, e): optional sign, then digits; stops at the first other character
inline int atoi_tok(const char *b, const char *e)
{
	bool neg = false;
	if (b < e && (*b == '-' || *b == '+'))
		neg = *b++ == '-';
	long long v = 0;
	while (b < e && *b >= '0' && *b <= '9')
		v = v * 10 + (*b++ - '0');
	return (int) (neg ? -v : v);
}

// atof on a token: plain decimals with <= 15 significant digits are exact as mantissa / 10^k (both exactly
// representable, one correctly rounded division = what strtod returns); anything else goes to strtod
inline double atof_tok(const char *b, const char *e)
{
	static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
	const char *p = b;
	bool neg = false;
	if (p < e && (*p == '-' || *p == '+'))
		neg = *p++ == '-';
	unsigned long long mant = 0;
	int digits = 0, frac = 0;
	bool seen_point = false, simple = p < e;
	for (; p < e; p++) {
		if (*p >= '0' && *p <= '9') {
			mant = mant * 10 + (unsigned) (*p - '0');
			digits++;
			if (seen_point)
				frac++;
		} else if (*p == '.' && !seen_point)
			seen_point = true;
		else {
			simple = false;
			break;
		}
	}
	if (simple && digits > 0 && digits <= 15 && frac <= 15) {
		const double v = (double) mant / p10[frac];
		return neg ? -v : v;
	}
	char tmp[64];
	const size_t n = std::min<size_t>((size_t) (e - b), sizeof tmp - 1);
	memcpy(tmp, b, n);
	tmp[n] = 0;
	return atof(tmp);
}

struct parsed_chunk {
	// rows of one byte range of the file, in file order; chromosome names are interned per chunk
	std::vector<std::string> names;
	std::vector<int32_t> name_id;
	std::vector<sv_row> row;
	std::vector<float> value;
};

void parse_range(const char *b, const char *e, bool with_value, parsed_chunk *out)
{
	const size_t guess = (size_t) (e - b) / 16 + 16; // ~25 bytes per row
	out->name_id.reserve(guess);
	out->row.reserve(guess);
	if (with_value)
		out->value.reserve(guess);
	int last_id = -1;
	const char *last_b = nullptr;
	size_t last_len = 0;
	while (b < e) {
		const char *eol = (const char *) memchr(b, '\n', (size_t) (e - b));
		const char *le = eol ? eol : e;
		const char *tok_b[4], *tok_e[4];
		int nt = 0;
		const char *p = b;
		while (p < le && nt < 4) {
			while (p < le && is_delim(*p))
				p++;
			if (p >= le)
				break;
			tok_b[nt] = p;
			while (p < le && !is_delim(*p))
				p++;
			tok_e[nt++] = p;
		}
		if (nt >= (with_value ? 4 : 3)) {
			const size_t len = (size_t) (tok_e[0] - tok_b[0]);
			if (last_id < 0 || len != last_len || memcmp(tok_b[0], last_b, len) != 0) {
				const std::string name(tok_b[0], len);
				last_id = -1;
				for (size_t k = 0; k < out->names.size(); k++)
					if (out->names[k] == name)
						last_id = (int) k;
				if (last_id < 0) {
					out->names.push_back(name);
					last_id = (int) out->names.size() - 1;
				}
				last_b = tok_b[0];
				last_len = len;
			}
			out->name_id.push_back(last_id);
			out->row.push_back(sv_row{atoi_tok(tok_b[1], tok_e[1]), atoi_tok(tok_b[2], tok_e[2])});
			if (with_value)
				out->value.push_back((float) atof_tok(tok_b[3], tok_e[3]));
		}
		b = eol ? eol + 1 : e;
	}
}

} // namespace

// Fast path: the file is mapped, cut at line ends into one range per thread and parsed without libc tokenisers;
// ranges are merged in file order, so every chromosome's rows keep their order.  Same rows as load_bed_fgets.
bool load_bed(const std::string &path, bool with_value, bed_index *out)
{
	if (knobs().bed_literal) // test hook: force the reference-literal reader
		return load_bed_fgets(path, with_value, out);
	const int fd = open(path.c_str(), O_RDONLY);
	if (fd < 0)
		return false;
	struct stat st;
	if (fstat(fd, &st) != 0 || st.st_size == 0) {
		close(fd);
		return st.st_size == 0 ? load_bed_fgets(path, with_value, out) : false;
	}
	const size_t size = (size_t) st.st_size;
	void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (m == MAP_FAILED)
		return load_bed_fgets(path, with_value, out);
	const char *data = (const char *) m;
	// a line of 511+ characters would be split by the reference's fgets(512): leave those files to the literal reader.
	// A NUL byte would end the reference's C strings early: same.
	bool plain = memchr(data, 0, size) == nullptr;
	for (const char *p = data; plain && p < data + size;) {
		const char *eol = (const char *) memchr(p, '\n', (size_t) (data + size - p));
		const size_t len = (size_t) ((eol ? eol : data + size) - p);
		if (len >= (size_t) kLineChunk - 1)
			plain = false;
		p = eol ? eol + 1 : data + size;
	}
	if (!plain) {
		munmap(m, size);
		return load_bed_fgets(path, with_value, out);
	}
	int n_threads = (int) std::thread::hardware_concurrency();
	n_threads = std::max(1, std::min(n_threads, 16));
	if (size < (size_t) 4 << 20)
		n_threads = 1;
	std::vector<const char *> cut(n_threads + 1);
	cut[0] = data;
	cut[n_threads] = data + size;
	for (int t = 1; t < n_threads; t++) {
		const char *p = data + size / n_threads * t;
		const char *eol = (const char *) memchr(p, '\n', (size_t) (data + size - p));
		cut[t] = eol ? eol + 1 : data + size;
	}
	std::vector<parsed_chunk> chunks(n_threads);
	std::vector<std::thread> pool;
	for (int t = 1; t < n_threads; t++)
		pool.emplace_back(parse_range, cut[t], cut[t + 1], with_value, &chunks[t]);
	parse_range(cut[0], cut[1], with_value, &chunks[0]);
	for (auto &th : pool)
		th.join();
	munmap(m, size);
	for (const parsed_chunk &c : chunks) {
		std::vector<std::vector<sv_row> *> rows(c.names.size());
		std::vector<std::vector<float> *> vals(c.names.size(), nullptr);
		for (size_t k = 0; k < c.names.size(); k++) {
			rows[k] = &out->rows[c.names[k]];
			if (with_value)
				vals[k] = &out->values[c.names[k]];
		}
		// append runs of rows of the same chromosome in bulk
		for (size_t i = 0; i < c.name_id.size();) {
			size_t j = i + 1;
			while (j < c.name_id.size() && c.name_id[j] == c.name_id[i])
				j++;
			rows[c.name_id[i]]->insert(rows[c.name_id[i]]->end(), c.row.begin() + (long) i, c.row.begin() + (long) j);
			if (with_value)
				vals[c.name_id[i]]->insert(vals[c.name_id[i]]->end(), c.value.begin() + (long) i, c.value.begin() + (long) j);
			i = j;
		}
	}
	return true;
}

std::vector<sv_row> known_SVs_for(const bed_index &bed, const std::string &chr, int min_sv_size)
{
	std::vector<sv_row> kept;
	auto it = bed.rows.find(chr);
	if (it == bed.rows.end())
		return kept;
	for (const sv_row &r : it->second)
		if (r.end - r.start >= min_sv_size)
			kept.push_back(r);
	// ascending (start, end).  Rows that compare equal are identical in every field the path uses,
	// so qsort's unspecified order among them cannot change the output.
	std::stable_sort(kept.begin(), kept.end(), [](const sv_row &a, const sv_row &b) {
		return a.start != b.start ? a.start < b.start : a.end < b.end;
	});
	return kept;
}

} // namespace conga_host
