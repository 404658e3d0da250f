// gpu_inflate_wave.hip -- PROTOTYPE of the next step: one BGZF block per WAVE.  All 64 lanes run
// the block decoder of conga_amd/host/inflate_core.h in lockstep on the same stream (tables in LDS, no divergence);
// a literal is stored by lane 0, a match is copied by all lanes.  Derived from
// gpu_inflate.hip -- stand-alone benchmark of the idea behind conga_reads_bgzf's first stage (the product version is
// bgzf_inflate_kernel in conga_amd/csrc/kernels_bam.hip.h): how fast does an MI355X inflate a BAM if every lane simply
// runs the host's block decoder (conga_amd/host/inflate_core.h, the same source) on its own BGZF block?
// No wave cooperation, tables in global scratch, byte stores.  Every inflated block is checked against its CRC32 on
// the host.  Measured: 2.9 GB/s inflated with 4 665 blocks in flight, 17.6 GB/s with 44 725.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpu_inflate tools/gpu_inflate.hip -lz && ./gpu_inflate file.bam
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../conga_amd/host/inflate_core.h"


#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

using namespace conga_host::inflate_core;

// the bytes this wave has written are read back by its match copies: loads that bypass the L1 (stores go through to L2)
__device__ __forceinline__ uint8_t load_written(const uint8_t *p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct WaveSymbols {
	__device__ static inline int run(Bits &b, const uint32_t *lit, const uint32_t *dist, uint8_t *out, uint8_t *&op, uint8_t *oend)
	{
		const int lane = threadIdx.x & 63;
		for (;;) {
			if (b.cnt < 0)
				return -1;
			b.refill();
			uint32_t e = decode(b, lit, kLitBits);
			if (e & kLiteral) { // (the same for every lane: no divergence)
				if (op == oend)
					return -1;
				if (lane == 0)
					*op = (uint8_t) (e >> 13);
				op++;
				continue;
			}
			if (!(e & kValid))
				return -1;
			if (e & kEnd)
				return 0;
			const size_t length = ((e >> 13) & 0x7FFFu) + b.take((int) ((e >> 8) & 31u));
			e = decode(b, dist, kDistBits);
			if (!(e & kValid))
				return -1;
			const size_t offset = ((e >> 13) & 0x7FFFu) + b.take((int) ((e >> 8) & 31u));
			if (b.cnt < 0 || offset > (size_t) (op - out) || length > (size_t) (oend - op))
				return -1;
			// everything this wave stored so far has to be in L2 before the copy reads it there: wait for the stores'
			// acknowledgements (s_waitcnt vmcnt(0); a release fence at agent scope also writes L2 back -- microseconds)
			__builtin_amdgcn_s_waitcnt(0x0F70);
			const uint8_t *src = op - offset;
			for (size_t k = (size_t) lane; k < length; k += 64)
				op[k] = load_written(src + (offset >= length ? k : k % offset));
			op += length;
		}
	}
	__device__ static inline void stored(uint8_t *out, uint8_t *op, const uint8_t *from, uint32_t len)
	{
		(void) out;
		for (uint32_t k = threadIdx.x & 63; k < len; k += 64)
			op[k] = from[k];
	}
};

__global__ __launch_bounds__(64) void inflate_kernel(int n_blocks, const uint8_t *in, const uint64_t *c_off, const uint32_t *c_len,
		uint8_t *out, const uint64_t *o_off, const uint32_t *o_len, Decoder *scratch, uint8_t *ok)
{
	__shared__ __attribute__((aligned(16))) unsigned char dec_raw[sizeof(Decoder)]; // the wave's tables
	Decoder &dec = *reinterpret_cast<Decoder *>(dec_raw);
	(void) scratch;
	for (int t = blockIdx.x; t < n_blocks; t += gridDim.x) {
		if (threadIdx.x == 0)
			dec.fixed_ready = false;
		__builtin_amdgcn_wave_barrier();
		const bool good = inflate_block_stream_t<WaveSymbols>(dec, in + c_off[t], c_len[t], out + o_off[t], o_len[t]);
		if (threadIdx.x == 0)
			ok[t] = good ? 1 : 0;
		__builtin_amdgcn_wave_barrier();
	}
}

int main(int argc, char **argv)
{
	if (argc < 2) {
		fprintf(stderr, "usage: gpu_inflate file.bam [max_blocks]\n");
		return 2;
	}
	FILE *f = fopen(argv[1], "rb");
	if (!f) {
		perror(argv[1]);
		return 1;
	}
	fseek(f, 0, SEEK_END);
	const size_t size = (size_t) ftell(f);
	fseek(f, 0, SEEK_SET);
	std::vector<uint8_t> file(size);
	if (fread(file.data(), 1, size, f) != size)
		return 1;
	fclose(f);
	const size_t max_blocks = argc > 2 ? (size_t) atol(argv[2]) : (size_t) 1 << 30;
	std::vector<uint64_t> c_off, o_off;
	std::vector<uint32_t> c_len, o_len, crc;
	uint64_t out_total = 0;
	for (size_t at = 0; at + 18 <= size && c_off.size() < max_blocks;) { // gzip member: 10 + XLEN(2) + extra + cdata + CRC32 + ISIZE
		const unsigned xlen = file[at + 10] | (file[at + 11] << 8);
		int bsize = -1;
		for (unsigned i = 0; i + 4 <= xlen;) {
			const uint8_t *x = &file[at + 12 + i];
			const unsigned slen = x[2] | (x[3] << 8);
			if (x[0] == 'B' && x[1] == 'C' && slen == 2)
				bsize = x[4] | (x[5] << 8);
			i += 4 + slen;
		}
		if (bsize < 0)
			break;
		const size_t cdata = (size_t) bsize + 1 - 12 - xlen - 8;
		uint32_t isize, c;
		memcpy(&c, &file[at + 12 + xlen + cdata], 4);
		memcpy(&isize, &file[at + 12 + xlen + cdata + 4], 4);
		if (isize) {
			c_off.push_back(at + 12 + xlen);
			c_len.push_back((uint32_t) cdata);
			o_off.push_back(out_total);
			o_len.push_back(isize);
			crc.push_back(c);
			out_total += isize;
		}
		at += (size_t) bsize + 1;
	}
	const int n = (int) c_off.size();
	printf("%s: %d blocks, %.1f MB compressed, %.1f MB inflated\n", argv[1], n, size / 1e6, out_total / 1e6);

	uint8_t *d_in, *d_out, *d_ok;
	uint64_t *d_c_off, *d_o_off;
	uint32_t *d_c_len, *d_o_len;
	Decoder *d_scratch;
	CHECK(hipMalloc(&d_in, size + 64)); // (the decoder reads a few aligned words ahead)
	CHECK(hipMalloc(&d_out, out_total + 16));
	CHECK(hipMalloc(&d_ok, (size_t) n));
	CHECK(hipMalloc(&d_c_off, (size_t) n * 8));
	CHECK(hipMalloc(&d_o_off, (size_t) n * 8));
	CHECK(hipMalloc(&d_c_len, (size_t) n * 4));
	CHECK(hipMalloc(&d_o_len, (size_t) n * 4));
	CHECK(hipMalloc(&d_scratch, sizeof(Decoder)));
	printf("tables: %.1f KB of LDS per wave\n", sizeof(Decoder) / 1e3);
	auto t0 = std::chrono::steady_clock::now();
	CHECK(hipMemcpy(d_in, file.data(), size, hipMemcpyHostToDevice));
	auto t1 = std::chrono::steady_clock::now();
	CHECK(hipMemcpy(d_c_off, c_off.data(), (size_t) n * 8, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_o_off, o_off.data(), (size_t) n * 8, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_c_len, c_len.data(), (size_t) n * 4, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(d_o_len, o_len.data(), (size_t) n * 4, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 3; rep++) {
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(inflate_kernel, dim3(n < 256 * 9 ? n : 256 * 9), dim3(64), 0, 0, n, d_in, d_c_off, d_c_len, d_out, d_o_off, d_o_len, d_scratch, d_ok);
		CHECK(hipEventRecord(e1));
		CHECK(hipEventSynchronize(e1));
		float ms;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("  inflate kernel: %.2f ms = %.1f GB/s compressed in, %.1f GB/s inflated out\n", ms, size / ms / 1e6, out_total / ms / 1e6);
		best = ms < best ? ms : best;
	}
	std::vector<uint8_t> out(out_total), ok((size_t) n);
	CHECK(hipMemcpy(out.data(), d_out, out_total, hipMemcpyDeviceToHost));
	CHECK(hipMemcpy(ok.data(), d_ok, (size_t) n, hipMemcpyDeviceToHost));
	long refused = 0, wrong = 0;
	auto t2 = std::chrono::steady_clock::now();
	for (int b = 0; b < n; b++) {
		if (!ok[(size_t) b])
			refused++;
		else if ((uint32_t) crc32(crc32(0L, Z_NULL, 0), out.data() + o_off[(size_t) b], o_len[(size_t) b]) != crc[(size_t) b])
			wrong++;
	}
	auto t3 = std::chrono::steady_clock::now();
	printf("checked against the blocks' CRC32: %ld refused, %ld wrong of %d\n", refused, wrong, n);
	printf("H2D of the compressed file: %.1f ms (pageable); CRC32 of %.1f MB on one host thread: %.0f ms\n",
			std::chrono::duration<double, std::milli>(t1 - t0).count(), out_total / 1e6, std::chrono::duration<double, std::milli>(t3 - t2).count());
	printf("RESULT best %.2f ms, %.1f GB/s inflated\n", best, out_total / best / 1e6);
	return (refused || wrong) ? 1 : 0;
}
