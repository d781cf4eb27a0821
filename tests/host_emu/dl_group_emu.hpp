// dl_group_emu.hpp -- TEST INFRASTRUCTURE: a host stand-in for the wave-level machinery the 16-lanes-per-walker
// kernels (drloco_amd/csrc/dl_group.hpp, dl_group_env.hpp) are written against, so that the SAME kernel source
// runs on a machine without a GPU and can be checked against the independent oracle there.
//
// A 64-lane wave is 64 cooperative fibers on one OS thread (a dozen lines of x86-64 context switch: glibc's swapcontext
// makes a system call per switch, which is what the run time would be spent on).  Every cross-lane operation of the kernels
// (DPP reads, ballots, the LDS exchange points g_sync) is a rendezvous: a lane publishes its operand, yields, and
// reads the other lanes' operands after every lane has arrived.  DPP reads and the LDS exchange points are row-local
// (the four walkers of a wave may be in different trips of a data-dependent loop, as under the GPU's EXEC mask), ballots
// are wave-wide: a row that reaches one waits for the others.  The scheduler checks that the 16 lanes of a row always
// arrive at the SAME kind of operation -- on the GPU a divergent cross-lane operation is silent garbage, here it aborts.
// Only what dl_group*.hpp uses is provided; float runs through the generic (non inline-asm) forms of the row
// primitives, so the emulation checks the algorithm, not the bit pattern of v_fmac_f32_dpp.
#pragma once

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <functional>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(x)

namespace dlemu {

constexpr int WAVE = 64;
constexpr size_t STACK = 1u << 20;

#if !defined(__x86_64__)
#error "dl_group_emu.hpp: the fiber switch is written for x86-64"
#endif
// save the callee-saved registers on the current stack, store its pointer in *save_sp, continue on new_sp
extern "C" void dlemu_swap(void** save_sp, void* new_sp);
asm(".text\n.globl dlemu_swap\n.type dlemu_swap,@function\ndlemu_swap:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size dlemu_swap,.-dlemu_swap\n");

struct Wave {
    void* main_sp = nullptr;
    void* sp[WAVE];
    std::vector<char> stacks;
    bool done[WAVE];
    int cur = 0, nlanes = WAVE;
    uint32_t slot[2][WAVE];      // double buffered: a lane publishes exchange k + 2 only after every lane of its row has read exchange k
    int parity[WAVE];
    int tag[WAVE];
    bool pred[WAVE];
    uint64_t ballot_result = 0;
    long ballot_gen = 0;
    long syncs = 0;
    std::function<void(int)> body;
};
constexpr int TAG_BALLOT = 1000;

inline Wave*& wave() { static thread_local Wave* w = nullptr; return w; }
inline int lane() { return wave()->cur; }

// rendezvous; `tag` names the kind of operation
inline void sync(int tag) {
    Wave* w = wave();
    w->tag[w->cur] = tag;
    dlemu_swap(&w->sp[w->cur], w->main_sp);
}

inline void trampoline() {
    Wave* w = wave();
    const int l = w->cur;
    w->body(l);
    w->done[l] = true;
    dlemu_swap(&w->sp[l], w->main_sp);
    abort();                              // a finished lane is never resumed
}

// prepare the fibers of a wave (nothing runs yet)
inline void wave_begin(Wave& w, int nlanes, const std::function<void(int)>& body) {
    w.nlanes = nlanes;
    w.body = body;
    w.stacks.resize(STACK * (size_t)nlanes);
    for (int l = 0; l < nlanes; l++) {
        w.done[l] = false; w.tag[l] = -1; w.slot[0][l] = w.slot[1][l] = 0; w.parity[l] = 0; w.pred[l] = false;
        // a fresh stack as dlemu_swap expects it: six register slots, then the entry point as return address
        uintptr_t top = ((uintptr_t)(w.stacks.data() + STACK * (size_t)(l + 1))) & ~(uintptr_t)15;
        void** f = (void**)(top - 64);
        for (int k = 0; k < 6; k++) f[k] = nullptr;
        f[6] = (void*)(void (*)())trampoline;
        f[7] = nullptr;
        w.sp[l] = (void*)f;
    }
}
// one round of a wave: every live lane runs to its next rendezvous; the rows' lockstep is checked, a complete ballot is resolved.  Returns false once every lane has
// finished.  `sleeping` (if given) tells whether the wave ended the round with every live lane in a poll's sleep (TAG_SLEEP): it cannot make progress by itself.
constexpr int TAG_SLEEP = 990, TAG_YIELD = 991;
inline bool wave_round(Wave& w, bool* sleeping = nullptr, bool* posted = nullptr) {
    Wave* prev = wave();
    wave() = &w;
    const int nlanes = w.nlanes;
    int live = 0, waiting = 0, asleep = 0, yielded = 0;
    for (int l = 0; l < nlanes; l++) {
        if (w.done[l]) continue;
        w.cur = l;
        dlemu_swap(&w.main_sp, w.sp[l]);
        if (w.done[l]) continue;
        live++;
        if (w.tag[l] == TAG_BALLOT) waiting++;
        if (w.tag[l] == TAG_SLEEP) asleep++;
        if (w.tag[l] == TAG_YIELD) yielded++;
    }
    wave() = prev;
    if (sleeping) *sleeping = live > 0 && asleep == live;
    if (posted) *posted = live > 0 && yielded == live;          // the wave has just posted a flag (DL_WAKE follows every flag store): its partner's poll may succeed now
    if (!live) return false;
    // the 16 lanes of a row move in lockstep
    for (int r = 0; r < nlanes; r += 16) {
        int tag0 = -2;
        for (int l = r; l < r + 16 && l < nlanes; l++) {
            const int t = w.done[l] ? -3 : w.tag[l];
            if (tag0 == -2) tag0 = t;
            else if (t != tag0) { fprintf(stderr, "dlemu: the lanes of a row diverged at a cross-lane operation (lane %d: op %d, lane %d: op %d)\n", r, tag0, l, t); abort(); }
        }
    }
    // a wave-wide ballot completes once every live lane has arrived
    if (waiting == live) {
        uint64_t m = 0;
        for (int l = 0; l < nlanes; l++) if (!w.done[l] && w.pred[l]) m |= 1ull << l;
        w.ballot_result = m;
        w.ballot_gen++;
    }
    w.syncs++;
    return true;
}
// run body(lane) for lanes 0..nlanes-1 as one wave
inline void run_wave(int nlanes, const std::function<void(int)>& body) {
    Wave w;
    wave_begin(w, nlanes, body);
    while (wave_round(w)) {}
}

// TWO waves that share memory (the dynamics wave and the partner wave of a split workgroup, tests/host_emu/emu.cpp): rounds of the two are interleaved by `policy`:
//   0  seeded random: each round goes to wave A with probability 1/2 (xorshift on `seed`)
//   1  A first: A runs whenever it is not asleep in a poll; B only while A sleeps          (the dynamics wave races ahead of its partner)
//   2  B first: B runs whenever it is not asleep; A only while B sleeps                      (the partner races ahead)
//   3  seeded random bursts: a wave keeps the SIMD for 1 .. 64 rounds
// Every cross-lane operation of the kernel source is a round boundary, so a hand-over protocol sees its partner's stores at thousands of different points of its own
// execution.  A wave asleep in a poll (DL_SLEEP) yields; two sleeping waves that stay asleep for `stall_limit` consecutive rounds are a deadlock: abort.
inline void run_pair(int nlanes, const std::function<void(int)>& body_a, const std::function<void(int)>& body_b, int policy, uint64_t seed, long stall_limit = 4000000) {
    Wave a, b;
    wave_begin(a, nlanes, body_a);
    wave_begin(b, nlanes, body_b);
    bool alive_a = true, alive_b = true, sleep_a = false, sleep_b = false;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    long stall = 0, burst = 0;
    bool cur_a = true;
    while (alive_a || alive_b) {
        bool pick_a;
        if (!alive_b) pick_a = true;
        else if (!alive_a) pick_a = false;
        else if (policy == 1) pick_a = !sleep_a || sleep_b;
        else if (policy == 2) pick_a = sleep_b && !sleep_a;
        else if (policy == 3) { if (burst <= 0) { cur_a = (rnd() >> 33) & 1; burst = 1 + (long)((rnd() >> 20) & 63); } burst--; pick_a = cur_a; }
        else pick_a = (rnd() >> 33) & 1;
        if ((policy == 1 || policy == 2) && sleep_a && sleep_b) pick_a = (rnd() >> 33) & 1;      // both in a poll: either may be the one whose flag has been posted
        bool posted = false;
        if (pick_a) { alive_a = wave_round(a, &sleep_a, &posted); if (posted) sleep_b = false; } else { alive_b = wave_round(b, &sleep_b, &posted); if (posted) sleep_a = false; }
        if ((sleep_a || !alive_a) && (sleep_b || !alive_b) && (alive_a || alive_b)) { if (++stall > stall_limit) { fprintf(stderr, "dlemu: both waves of a pair are asleep in their polls: deadlock\n"); abort(); } }
        else stall = 0;
    }
}

// one 32-bit word per lane: publish, rendezvous, read lane `src` (or 0 if src < 0)
inline uint32_t exchange(uint32_t x, int src, int tag) {
    Wave* w = wave();
    const int l = w->cur, par = w->parity[l];
    w->parity[l] = par ^ 1;
    w->slot[par][l] = x;
    sync(tag);
    return (src >= 0 && src < w->nlanes) ? w->slot[par][src] : 0u;
}

inline int dpp_src(int l, int ctrl) {
    const int row = l & ~15, i = l & 15;
    if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = i + (ctrl - 0x100); return s < 16 ? row + s : -1; }          // row_shl:n  (lane i reads lane i + n)
    if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = i - (ctrl - 0x110); return s >= 0 ? row + s : -1; }          // row_shr:n  (lane i reads lane i - n)
    if (ctrl >= 0x121 && ctrl <= 0x12f) return row + ((i - (ctrl - 0x120)) & 15);                                    // row_ror:n
    if (ctrl >= 0x150 && ctrl <= 0x15f) return row + (ctrl - 0x150);                                                 // row_newbcast:k
    fprintf(stderr, "dlemu: DPP control 0x%x is not emulated\n", ctrl);
    abort();
}

inline uint64_t ballot(bool p) {
    Wave* w = wave();
    const int l = w->cur;
    w->pred[l] = p;
    const long gen = w->ballot_gen;
    do sync(TAG_BALLOT); while (w->ballot_gen == gen);
    const uint64_t m = w->ballot_result;
    sync(TAG_BALLOT + 1);               // nobody starts the next ballot before everybody has read this one
    return m;
}

struct Dim3 { unsigned x = 0, y = 0, z = 0; };

}  // namespace dlemu

// ---- the names the kernel source uses
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;      // the kernels only use full masks with old = 0 / bound_ctrl
    return (int)dlemu::exchange((uint32_t)src, dlemu::dpp_src(dlemu::lane(), ctrl), 2000 + 2 * ctrl);
}
inline void __builtin_amdgcn_wave_barrier() { dlemu::sync(900); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
inline uint64_t __ballot(bool p) { return dlemu::ballot(p); }
inline bool __any(bool p) { return dlemu::ballot(p) != 0; }
inline int __popc(uint32_t x) { return __builtin_popcount(x); }
inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
