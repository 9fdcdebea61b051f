// Stand-alone check of the register conflict table of k_tab_pwg<false> (find_conflicts in
// csrc/tabular_pwg.hip) against the definition: random batches, B lanes, a random set of writers.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/exp_regtable scripts/experiments/exp_regtable.hip && /tmp/exp_regtable
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int kBuckets = 32;
__device__ __forceinline__ uint32_t bucket_of(uint32_t state) {
  return ((state * 0x9E5u) >> 6) & (uint32_t)(kBuckets - 1);
}

// out[case][lane] = {bn, bs, table}; stop[case]
__global__ void k(const uint32_t* idx_in, const uint32_t* ns_in, const uint32_t* ch_in, int B, int first,
                  uint32_t* out, int* stop_out) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x;
  if (lane < B) {
    const uint32_t idx = idx_in[c * 64 + lane], ns = ns_in[c * 64 + lane];
    const bool act = lane >= first;
    const bool ch = act && ch_in[c * 64 + lane] != 0;
    const uint32_t slot_s = bucket_of(idx >> 2) * 4u, slot_n = bucket_of(ns) * 4u;
    const uint32_t mine = 0x80000000u | (idx << 6) | (uint32_t)lane;
    const uint32_t table = (uint32_t)__builtin_amdgcn_ds_permute((int)(ch ? slot_s : (uint32_t)kBuckets * 4u), (int)mine);
    const uint32_t bn = (uint32_t)__builtin_amdgcn_ds_bpermute((int)slot_n, (int)table);
    const uint32_t bs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)slot_s, (int)table);
    const bool row_hit = ((bn & 0x7fffffffu) >> 8) == ns && (int)(bn & 63u) < lane && (int)bn < 0;
    const bool cell_hit = ((bs & 0x7fffffffu) >> 6) == idx && (int)(bs & 63u) < lane && (int)bs < 0;
    const bool lost = ch && bs != mine;
    const unsigned long long wait = __builtin_amdgcn_ballot_w64(act && (row_hit || cell_hit || lost));
    int stop = B;
    if (wait) stop = ((wait >> first) & 1ull) ? first + 1 : __ffsll((long long)wait) - 1;
    out[(c * 64 + lane) * 3 + 0] = bn;
    out[(c * 64 + lane) * 3 + 1] = bs;
    out[(c * 64 + lane) * 3 + 2] = table;
    if (lane == 0) stop_out[c] = stop;
  }
}

int main() {
  const int cases = 4096, B = 50;
  uint32_t *idx, *ns, *ch, *out;
  int* stop;
  hipMallocManaged(&idx, cases * 64 * 4);
  hipMallocManaged(&ns, cases * 64 * 4);
  hipMallocManaged(&ch, cases * 64 * 4);
  hipMallocManaged(&out, cases * 64 * 12);
  hipMallocManaged(&stop, cases * 4);
  srand(1);
  for (int i = 0; i < cases * 64; ++i) {
    idx[i] = rand() % 4096;
    ns[i] = rand() % 1024;
    ch[i] = (rand() % 50) < (i / 64 % 12);
  }
  int bad = 0, exact = 0, early = 0, lostc = 0;
  for (int first = 0; first < 3; ++first) {
    k<<<cases, 64>>>(idx, ns, ch, B, first * 7, out, stop);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    for (int c = 0; c < cases; ++c) {
      const int f = first * 7;
      // definition: lowest lane j >= f with an earlier writer k in [f, j) that writes a cell j reads
      int want = B;
      for (int j = f; j < B && want == B; ++j)
        for (int kk = f; kk < j; ++kk)
          if (ch[c * 64 + kk] && (idx[c * 64 + kk] == idx[c * 64 + j] || (idx[c * 64 + kk] >> 2) == ns[c * 64 + j])) { want = j; break; }
      const int got = stop[c];
      if (got > want) { if (++bad < 10) printf("case %d first %d: stop %d > exact %d\n", c, f, got, want); }
      else if (got == want) ++exact;
      else ++early;
      // every writer finds itself in its bucket unless two writers share one
      for (int j = f; j < B; ++j)
        if (ch[c * 64 + j] && out[(c * 64 + j) * 3 + 1] != (0x80000000u | (idx[c * 64 + j] << 6) | j)) ++lostc;
    }
  }
  printf("cases %d: exact %d, earlier than needed %d (bucket collisions), WRONG %d; writers that lost a bucket %d\n",
         cases * 3, exact, early, bad, lostc);
  return bad != 0;
}
