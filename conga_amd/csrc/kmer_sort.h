// kmer_sort.h -- see kmer_sort.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace conga {

// positions_out[] = 0 .. n - 1 ordered by keys_in[] (ties: ascending), keys_out[] = the keys in that order; only the low
// `key_bits` bits of a key count.  temp == nullptr: *temp_bytes receives the scratch size and nothing runs.  -> hipError_t
int kmer_sort_positions(void *temp, size_t *temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, int32_t *positions_out, uint32_t n,
		unsigned key_bits, hipStream_t stream);

} // namespace conga
