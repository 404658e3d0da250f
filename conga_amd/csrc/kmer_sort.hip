// kmer_sort.hip -- the one library call of the engine: a stable LSD radix sort (rocPRIM, AMD's own primitives) that
// orders a chromosome's positions by their 10-mer.  build_hash_table (split_read.c:394-440) appends every position to its
// 10-mer's bucket in increasing order; a STABLE sort of 0, 1, 2, ... by the 10-mer's hash gives exactly those buckets, one
// behind the other.  Runs once per layout (the index depends on the reference sequence only), never in a sample's step.
// A translation unit of its own: the templates take 13 s to compile, conga_api.hip is rebuilt far more often.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "kmer_sort.h"

namespace conga {

int kmer_sort_positions(void *temp, size_t *temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, int32_t *positions_out, uint32_t n,
		unsigned key_bits, hipStream_t stream)
{
	size_t bytes = *temp_bytes;
	rocprim::counting_iterator<int32_t> iota(0);
	const hipError_t e = rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, iota, positions_out, n, 0u, key_bits, stream);
	*temp_bytes = bytes;
	return (int) e;
}

} // namespace conga
