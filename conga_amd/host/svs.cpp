#include "svs.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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

bool load_bed(const std::string &path, bool with_value, bed_index *out)
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
