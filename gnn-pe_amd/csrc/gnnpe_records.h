// gnnpe_records.h -- layouts of the enumeration's device records (no kernels), shared by the translation units that
// read them: the emit kernels (gnnpe_engine.hip) and the pair-major index build (gnnpe_index.hip).
#pragma once

#include <cstdint>

namespace gnnpe {

constexpr uint32_t kNoEdge = 0xFFFFFFFFu;
constexpr uint32_t kHubFlag = 0x80000000u;  // RankedPair::cnt bit: the pair's middle row is a hub row
constexpr uint32_t kHubDegree = 64;         // rows longer than this are hub rows
constexpr uint32_t kRowAlign = 128;         // row blocks start on L2 lines: the fabric fetches whole 128-byte lines

// per (s, b) pair, indexed by the pair's emission index (poffs[rank[s]] + position of b in N(s)): 16 bytes, one
// dwordx4 store scattered by the row kernel, one contiguous load per lane in the emit kernel
struct __attribute__((aligned(16))) RankedPair {
    uint32_t block;  // row block of b, in kRowAlign units: header vde[b], then the records
    uint32_t cnt;    // paths of the pair = records to read from the front of the block (| kHubFlag: hub row)
    uint64_t G;      // id-positions of N(b) with rank > rank[s]; hub: deg(b)
};

// Neighbour records.  Wide: {id, id-position, vde} (hub rows: {id, rank, vde}); packed (graphs of up to 2^26
// vertices): the id-position rides in the id's top 6 bits -- 4 bytes less per emitted path to fetch.
template <int E> struct __attribute__((packed, aligned(4))) RecWide {
    uint32_t id, aux;
    double vde[E];
};
template <int E> struct __attribute__((packed, aligned(4))) RecPacked {
    uint32_t idp;
    double vde[E];
};
constexpr uint32_t kPackedIdBits = 26;
template <int E, bool PACKED> struct RecOf { typedef RecWide<E> type; };
template <int E> struct RecOf<E, true> { typedef RecPacked<E> type; };

// per start vertex of the slab: what the emit kernel needs before it touches the pairs
struct __attribute__((aligned(16))) StartRec {
    uint64_t base, end;  // first / one-past-last output slot of this start vertex
    uint32_t e0, ds, s, part;
    uint32_t a_s, pad0, pad1, pad2;  // adj_start[s]: N(s) = the middle vertices of its pairs, in pair order
};

}  // namespace gnnpe
