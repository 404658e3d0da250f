// inflate_prof.hip -- where a wave of bgzf_inflate_wave_kernel spends its time: the kernel of conga_amd/csrc/inflate_wave.hip.h
// built with IW_PROF (cycle counters around its phases), run over the blocks of a BAM file.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIW_PROF -o tools/inflate_prof tools/inflate_prof.hip && tools/inflate_prof file.bam [max_blocks]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../conga_amd/csrc/inflate_wave.hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv)
{
	if (argc < 2)
		return 2;
	FILE *f = fopen(argv[1], "rb");
	if (!f)
		return 1;
	fseek(f, 0, SEEK_END);
	const size_t size = (size_t) ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> file(size);
	if (fread(file.data(), 1, size, f) != size)
		return 1;
	fclose(f);
	const size_t max_blocks = argc > 2 ? (size_t) atol(argv[2]) : (size_t) 1 << 30;
	std::vector<conga_bgzf_block> blocks;
	std::vector<uint64_t> off;
	uint64_t total = 0;
	for (size_t at = 0; at + 18 <= size && blocks.size() < max_blocks;) {
		const size_t bsize = (size_t) (file[at + 16] | (file[at + 17] << 8)) + 1;
		conga_bgzf_block b;
		memset(&b, 0, sizeof b);
		b.data_off = at + 18;
		b.data_len = (uint32_t) (bsize - 26);
		memcpy(&b.crc32, &file[at + bsize - 8], 4);
		memcpy(&b.inflated_len, &file[at + bsize - 4], 4);
		if (b.inflated_len) {
			off.push_back(total);
			total += b.inflated_len;
			blocks.push_back(b);
		}
		at += bsize;
	}
	const uint32_t n = (uint32_t) blocks.size();
	uint32_t crc[256], x2n[32];
	for (uint32_t i = 0; i < 256; i++) {
		uint32_t c = i;
		for (int k = 0; k < 8; k++)
			c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
		crc[i] = c;
	}
	auto mul = [](uint32_t a, uint32_t b) {
		uint32_t p = 0;
		for (int k = 0; k < 32; k++) {
			if ((a >> (31 - k)) & 1u)
				p ^= b;
			b = (b & 1u) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
		}
		return p;
	};
	x2n[0] = 0x40000000u;
	for (int k = 1; k < 32; k++)
		x2n[k] = mul(x2n[k - 1], x2n[k - 1]);
	uint8_t *d_in, *d_out, *d_status;
	conga_bgzf_block *d_blocks;
	uint64_t *d_off;
	uint32_t *d_crc, *d_x2n;
	CHECK(hipMalloc(&d_in, size + 512));
	CHECK(hipMalloc(&d_out, total + 16));
	CHECK(hipMalloc(&d_status, n));
	CHECK(hipMalloc(&d_blocks, n * sizeof(conga_bgzf_block)));
	CHECK(hipMalloc(&d_off, n * 8));
	CHECK(hipMalloc(&d_crc, sizeof crc));
	CHECK(hipMalloc(&d_x2n, sizeof x2n));
	CHECK(hipMemcpy(d_in, file.data(), size, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_blocks, blocks.data(), n * sizeof(conga_bgzf_block), hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_off, off.data(), n * 8, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_crc, crc, sizeof crc, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_x2n, x2n, sizeof x2n, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	for (int rep = 0; rep < 3; rep++) {
#ifdef IW_PROF
		unsigned long long zero[conga::iw::P_N] = {};
		CHECK(hipMemcpyToSymbol(HIP_SYMBOL(conga::iw::g_prof), zero, sizeof zero));
#endif
		const unsigned groups = (unsigned) std::min<size_t>((n + 3) / 4, 256 * 8);
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(conga::iw::bgzf_inflate_wave_kernel<false>, dim3(groups), dim3(256), 0, 0, n, d_in, d_blocks, d_off, d_out, d_crc, d_x2n, d_status);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
#ifdef IW_PROF
		unsigned long long p[conga::iw::P_N];
		CHECK(hipMemcpyFromSymbol(p, HIP_SYMBOL(conga::iw::g_prof), sizeof p));
#endif
		std::vector<uint8_t> st(n);
		CHECK(hipMemcpy(st.data(), d_status, n, hipMemcpyDeviceToHost));
		size_t bad = 0;
		for (uint8_t s : st)
			bad += s != 0;
		const char *names[] = {"view+lookup", "walk", "literal stores", "match wait", "match copy", "tables", "crc"};
		printf("%u blocks, %.1f MB inflated, %.2f ms (%.1f GB/s), %zu not ok\n", n, total / 1e6, ms, total / ms / 1e6, bad);
#ifdef IW_PROF
		printf("  per block: %.0f trips, %.0f symbols (%.2f per trip), %.0f matches, %.0f store waits\n", (double) p[conga::iw::P_TRIPS] / n,
				(double) p[conga::iw::P_SYMS] / n, (double) p[conga::iw::P_SYMS] / (double) p[conga::iw::P_TRIPS], (double) p[conga::iw::P_MATCHES] / n,
				(double) p[conga::iw::P_WAITS] / n);
		printf("  per block: %.0f matches one after the other (%.0f longer than 8 bytes, %.0f overlapping themselves), %.0f bytes\n",
				(double) p[conga::iw::P_SLOW] / n, (double) p[conga::iw::P_SLOW_LONG] / n, (double) p[conga::iw::P_SLOW_OVERLAP] / n,
				(double) p[conga::iw::P_SLOW_BYTES] / n);
		printf("  per block: of those %.0f of 9..16 bytes, %.0f of 17..32, %.0f of 33..64; %.0f of 9..32 bytes that do not overlap themselves and whose source lies in front of the trip's output\n",
				(double) p[conga::iw::P_SLOW_LE16] / n, (double) p[conga::iw::P_SLOW_LE32] / n, (double) p[conga::iw::P_SLOW_LE64] / n, (double) p[conga::iw::P_SLOW_SAFE32] / n);
		unsigned long long sum = 0;
		for (int k = 0; k < 7; k++)
			sum += p[k];
		for (int k = 0; k < 7; k++)
			printf("  %-16s %10.0f ticks per block (%.1f %%)  %.0f per trip\n", names[k], (double) p[k] / n, 100.0 * p[k] / sum, (double) p[k] / (double) p[conga::iw::P_TRIPS]);
#endif
	}
	return 0;
}
