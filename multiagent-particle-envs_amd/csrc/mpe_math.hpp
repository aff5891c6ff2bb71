// Scalar building blocks of the particle-world step, shared by every kernel.
//
// The generic templates perform the same IEEE operations in the same order as the reference's numpy
// scalar code (compile with -ffp-contract=off so no multiply-add is fused); they are what the
// float64 kernels run.  The float specialisations are the throughput variant: same formulas on the
// hardware's 1-ulp v_rsq/v_rcp/v_sqrt/v_exp/v_log instructions with fused multiply-adds, good to
// float32 rounding (trajectory bar 1e-4 after 25 steps) but not bit-faithful.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpe_launch.hpp"

namespace mpe {

template <typename T> struct V2 { T x, y; };

// A global store, non-temporal when NT.  Plain stores leave their lines dirty in the L2 and a kernel's end writes every
// dirty line back, so a launch of ONE step -- whose waves all store in the last third of its life -- ends with its whole
// output (20 MB at 65 536 worlds) draining after the last wave has gone.  A non-temporal store is written through as it
// is issued: at 65 536 worlds of simple_spread a single step takes 5.1 us instead of 6.0 (one launch per step under a
// hipGraph).  Episode launches, which store for their whole life, are 3 % slower with them and keep plain stores.
template <bool NT, typename V>
__device__ __forceinline__ void put(V* dst, V v) {
    if constexpr (NT) __builtin_nontemporal_store(v, dst);
    else *dst = v;
}


template <typename T> __device__ __forceinline__ T t_sqrt(T v);
template <> __device__ __forceinline__ float t_sqrt<float>(float v) { return __builtin_amdgcn_sqrtf(v); }
template <> __device__ __forceinline__ double t_sqrt<double>(double v) { return sqrt(v); }

template <typename T> __device__ __forceinline__ T t_exp(T v);
template <> __device__ __forceinline__ float t_exp<float>(float v) { return __builtin_amdgcn_exp2f(v * 1.44269504088896340736f); }
template <> __device__ __forceinline__ double t_exp<double>(double v) { return exp(v); }

template <typename T> __device__ __forceinline__ T t_log1p(T v);
// v >= 0 here (v = exp(-|y|)); the absolute error of log(1 + v) stays below one float ulp of 1
template <> __device__ __forceinline__ float t_log1p<float>(float v) { return __builtin_amdgcn_logf(1.0f + v) * 0.693147180559945309417f; }
template <> __device__ __forceinline__ double t_log1p<double>(double v) { return log1p(v); }

template <typename T> __device__ __forceinline__ T t_abs(T v);
template <> __device__ __forceinline__ float t_abs<float>(float v) { return fabsf(v); }
template <> __device__ __forceinline__ double t_abs<double>(double v) { return fabs(v); }

// sqrt(sum(square(delta))) of a 2-vector: multiagent/core.py:224 and every is_collision.
template <typename T> __device__ __forceinline__ T norm2(T dx, T dy) { return t_sqrt<T>(dx * dx + dy * dy); }
template <> __device__ __forceinline__ float norm2<float>(float dx, float dy) { return __builtin_amdgcn_sqrtf(__builtin_fmaf(dx, dx, dy * dy)); }

// numpy.logaddexp(0, y) (npy_logaddexp): y == 0 -> ln 2; y < 0 -> log1p(exp(y));
// y > 0 -> y + log1p(exp(-y)); NaN propagates.  multiagent/core.py:229.
template <typename T> __device__ __forceinline__ T logaddexp0(T y) {
    const T ay = t_abs<T>(y);
    const T pos = y > T(0) ? y : T(0);
    T r = pos + t_log1p<T>(t_exp<T>(-ay));
    if (y == T(0)) r = T(0.693147180559945309417232121458176568);
    return r;
}

// Force on entity a of the pair (a, b); entity b receives the negation.
// World.get_collision_force, multiagent/core.py:222-230:
//   delta = pa - pb; dist = |delta|; pen = logaddexp(0, -(dist - dist_min)/k) * k;
//   force = contact_force * delta / dist * pen          (left to right)
// dist == 0 yields NaN exactly as the reference does.
template <typename T>
__device__ __forceinline__ V2<T> contact_force(V2<T> pa, V2<T> pb, T dist_min, T cf, T k) {
    const T dx = pa.x - pb.x, dy = pa.y - pb.y;
    const T dist = norm2<T>(dx, dy);
    const T pen = logaddexp0<T>(-(dist - dist_min) / k) * k;
    V2<T> f;
    f.x = cf * dx / dist * pen;
    f.y = cf * dy / dist * pen;
    return f;
}

// float32: one v_rsq gives both dist and 1/dist; the divisions by dist and by k become multiplies.
// dist == 0 -> rsq = inf -> dist = 0 * inf = NaN, so the NaN of the reference is kept.
template <>
__device__ __forceinline__ V2<float> contact_force<float>(V2<float> pa, V2<float> pb, float dist_min, float cf, float k) {
    const float dx = pa.x - pb.x, dy = pa.y - pb.y;
    const float d2 = __builtin_fmaf(dx, dx, dy * dy);
    const float inv = __builtin_amdgcn_rsqf(d2);
    const float dist = d2 * inv;
    const float y = (dist_min - dist) * __builtin_amdgcn_rcpf(k);
    const float e = __builtin_amdgcn_exp2f(-fabsf(y) * 1.44269504088896340736f);
    const float sp = fmaxf(y, 0.0f) + __builtin_amdgcn_logf(1.0f + e) * 0.693147180559945309417f;
    const float s = cf * inv * (sp * k);
    V2<float> f;
    f.x = dx * s;
    f.y = dy * s;
    return f;
}

// World.get_wall_collision_force, multiagent/core.py:270-307.  Returns false when the wall exerts
// no force (entity beyond the wall's ends, or a ghost agent at a soft wall).  float64 follows the
// reference's operation order; float32 uses the same formulas in float.
template <typename T>
__device__ __forceinline__ bool wall_force(V2<T> p, T size, bool ghost, bool vertical, bool hard, T axis_pos, T end0, T end1, T width,
                                           T cf, T k, V2<T>& out) {
    if (ghost && !hard) return false;
    const T prll = vertical ? p.y : p.x, perp = vertical ? p.x : p.y;
    if (prll < end0 - size || prll > end1 + size) return false;
    T theta = T(0), dist_min;
    if (prll < end0 || prll > end1) {               // part of the entity is beyond the wall's end
        const T past = (prll < end0) ? prll - end0 : prll - end1;
        theta = (T)asin((double)(past / size));
        dist_min = (T)cos((double)theta) * size + T(0.5) * width;
    } else {
        dist_min = size + T(0.5) * width;
    }
    const T delta = perp - axis_pos;
    const T dist = t_abs<T>(delta);
    const T pen = logaddexp0<T>(-(dist - dist_min) / k) * k;
    const T mag = cf * delta / dist * pen;
    const T f_perp = (T)cos((double)theta) * mag;
    const T f_prll = (T)sin((double)theta) * t_abs<T>(mag);
    out.x = vertical ? f_perp : f_prll;
    out.y = vertical ? f_prll : f_perp;
    return true;
}

// For kernels that test many (entity, wall) pairs: the usual case -- the entity lies between the
// wall's ends, theta = 0 (core.py:299-302) -- is evaluated inline without trigonometry
// (cos(0) * mag = mag, sin(0) * |mag| = 0 * |mag|); an entity sticking out past an end takes the
// out-of-line general routine, so the kernel carries one copy of it instead of one per pair.
// The result travels back in registers: a reference parameter of an out-of-line function is a scratch-memory
// slot, and reading it back makes the caller wait for every older memory operation of the wave, the previous
// step's output stores included.
template <typename T> struct WallHit { T x, y; int hit; };

template <typename T>
__device__ __noinline__ WallHit<T> wall_force_past_end(V2<T> p, T size, bool ghost, bool vertical, bool hard, T axis_pos, T end0, T end1, T width,
                                                       T cf, T k) {
    V2<T> f;
    f.x = T(0); f.y = T(0);
    WallHit<T> r;
    r.hit = wall_force<T>(p, size, ghost, vertical, hard, axis_pos, end0, end1, width, cf, k, f) ? 1 : 0;
    r.x = f.x; r.y = f.y;
    return r;
}

template <typename T>
__device__ __forceinline__ bool wall_force_fast(V2<T> p, T size, bool ghost, bool vertical, bool hard, T axis_pos, T end0, T end1, T width,
                                                T cf, T k, V2<T>& out) {
    if (ghost && !hard) return false;
    const T prll = vertical ? p.y : p.x, perp = vertical ? p.x : p.y;
    if (prll < end0 - size || prll > end1 + size) return false;
    if (prll < end0 || prll > end1) {
        const WallHit<T> r = wall_force_past_end<T>(p, size, ghost, vertical, hard, axis_pos, end0, end1, width, cf, k);
        out.x = r.x; out.y = r.y;
        return r.hit != 0;
    }
    const T dist_min = size + T(0.5) * width;
    const T delta = perp - axis_pos;
    const T dist = t_abs<T>(delta);
    const T pen = logaddexp0<T>(-(dist - dist_min) / k) * k;
    const T mag = cf * delta / dist * pen;
    const T f_prll = T(0) * t_abs<T>(mag);
    out.x = vertical ? mag : f_prll;
    out.y = vertical ? f_prll : mag;
    return true;
}

// Scenario.is_collision: strict dist < size_1 + size_2 (simple_spread.py:77-81,
// simple_tag.py:69-73, simple_world_comm.py:126-130).
template <typename T> __device__ __forceinline__ bool touching(V2<T> p1, V2<T> p2, T dist_min) {
    return norm2<T>(p1.x - p2.x, p1.y - p2.y) < dist_min;
}

// The radicand of norm2, rounded the same way.  sqrt is monotonic, so a minimum over distances can
// be taken over these and rooted once: min_a sqrt(s_a) == sqrt(min_a s_a) bit for bit.
template <typename T> __device__ __forceinline__ T sqdist(T dx, T dy) { return dx * dx + dy * dy; }
template <> __device__ __forceinline__ float sqdist<float>(float dx, float dy) { return __builtin_fmaf(dx, dx, dy * dy); }
template <typename T> __device__ __forceinline__ T root(T s) { return t_sqrt<T>(s); }
template <> __device__ __forceinline__ float root<float>(float s) { return __builtin_amdgcn_sqrtf(s); }

// touching() for loops over many pairs: the squared distance decides unless it lies within a few
// ulps of dist_min^2, where the reference's own comparison sqrt(s) < dist_min is evaluated.  Same
// truth value as touching() for every input (a NaN distance is "not touching").
template <typename T> __device__ __forceinline__ bool touching_many(V2<T> p1, V2<T> p2, T dist_min) {
    const T s = sqdist<T>(p1.x - p2.x, p1.y - p2.y);
    const T d2 = dist_min * dist_min;
    const T slack = T(32) * (sizeof(T) == 4 ? T(1.1920929e-7) : T(2.220446049250313e-16));
    if (s < d2 * (T(1) - slack)) return true;
    if (!(s < d2 * (T(1) + slack))) return false;
    return root<T>(s) < dist_min;
}

// bound(x) of the boundary penalty, simple_tag.py:103-108 / simple_world_comm.py:167-172.
template <typename T> __device__ __forceinline__ T arena_bound(T x) {
    if (x < T(0.9)) return T(0);
    if (x < T(1.0)) return (x - T(0.9)) * T(10);
    const T e = t_exp<T>(T(2) * x - T(2));
    return T(10) < e ? T(10) : e;   // Python min(e, 10): keeps e unless 10 < e (a NaN stays)
}

// Python's min(list) on floats: keeps the first element unless a later one compares smaller.
template <typename T> __device__ __forceinline__ T py_min(T cur, T item) { return item < cur ? item : cur; }

// One entity's World.integrate_state, multiagent/core.py:196-204.
template <typename T, bool CLAMP>
__device__ __forceinline__ void integrate(V2<T>& p, V2<T>& v, V2<T> force, T mass, T max_speed, T one_minus_damping, T dt) {
    v.x = v.x * one_minus_damping;
    v.y = v.y * one_minus_damping;
    if (mass != T(1)) { force.x = force.x / mass; force.y = force.y / mass; }   // x / 1.0 == x
    v.x = v.x + force.x * dt;
    v.y = v.y + force.y * dt;
    if (CLAMP) {
        const T speed = t_sqrt<T>(v.x * v.x + v.y * v.y);
        if (speed > max_speed) {
            v.x = v.x / speed * max_speed;
            v.y = v.y / speed * max_speed;
        }
    }
    p.x = p.x + v.x * dt;
    p.y = p.y + v.y * dt;
}

template <bool CLAMP>
__device__ __forceinline__ void integrate_f32(V2<float>& p, V2<float>& v, V2<float> force, float mass, float max_speed, float one_minus_damping, float dt) {
    if (mass != 1.0f) { const float im = __builtin_amdgcn_rcpf(mass); force.x *= im; force.y *= im; }
    v.x = __builtin_fmaf(force.x, dt, v.x * one_minus_damping);
    v.y = __builtin_fmaf(force.y, dt, v.y * one_minus_damping);
    if (CLAMP) {
        const float s2 = __builtin_fmaf(v.x, v.x, v.y * v.y);
        if (s2 > max_speed * max_speed) {
            const float sc = max_speed * __builtin_amdgcn_rsqf(s2);
            v.x *= sc;
            v.y *= sc;
        }
    }
    p.x = __builtin_fmaf(v.x, dt, p.x);
    p.y = __builtin_fmaf(v.y, dt, p.y);
}
template <> __device__ __forceinline__ void integrate<float, true>(V2<float>& p, V2<float>& v, V2<float> f, float m, float ms, float omd, float dt) { integrate_f32<true>(p, v, f, m, ms, omd, dt); }
template <> __device__ __forceinline__ void integrate<float, false>(V2<float>& p, V2<float>& v, V2<float> f, float m, float ms, float omd, float dt) { integrate_f32<false>(p, v, f, m, ms, omd, dt); }

// MultiAgentEnv._set_action for a discrete int move, multiagent/environment.py:176-181,192-195.
template <typename T> __device__ __forceinline__ V2<T> decode_move(int a, T sensitivity) {
    V2<T> u;
    u.x = (a == 1) ? T(-1) : ((a == 2) ? T(1) : T(0));
    u.y = (a == 3) ? T(-1) : ((a == 4) ? T(1) : T(0));
    u.x = u.x * sensitivity;
    u.y = u.y * sensitivity;
    return u;
}

// ---- XcdDeal, device side (see mpe_launch.hpp) ----
__device__ __forceinline__ unsigned hw_xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), bits 2:0: simm16 = (size - 1) << 11 | offset << 6 | id
    return __builtin_amdgcn_s_getreg((2 << 11) | (0 << 6) | 20) & 7u;
}
// The unit workgroup `wg` takes, or -1 (uniform over the workgroup: every thread may call it).
__device__ __forceinline__ long long xcd_unit(const XcdDeal& d, unsigned wg) {
    const unsigned s = (unsigned)*d.phase & 7u;
    const unsigned j = wg + s, x = j & 7u, slot = (j >> 3) - (x < s ? 1u : 0u);
    if (slot >= (unsigned)d.share[x & 1u]) return -1;
    const long long unit = (long long)(x >> 1) * (d.share[0] + d.share[1]) + (x & 1u) * d.share[0] + slot;
    return unit < d.units ? unit : -1;
}

// ---------------------------------------------------------------------------------------------
// Counter-based uniform generator for on-device reset: Philox-4x32-10.
// ---------------------------------------------------------------------------------------------
struct Philox {
    __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __host__ __device__ static inline void block(uint64_t key, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
        uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
        out[0] = (uint32_t)ctr_lo; out[1] = (uint32_t)(ctr_lo >> 32);
        out[2] = (uint32_t)ctr_hi; out[3] = (uint32_t)(ctr_hi >> 32);
        for (int r = 0; r < 10; ++r) {
            round(out, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    // 53-bit uniform in [0, 1) from two 32-bit words (same construction as numpy's random_sample).
    __host__ __device__ static inline double u01(uint32_t a, uint32_t b) {
        return (double)(((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6)) * (1.0 / 9007199254740992.0);
    }
};

// Position of entity `e` of world `b` in episode rs.episode: the two coordinates come from the
// Philox block with key = seed and counter = (rs.env_base + b, episode * 256 + e) -- the world's index in
// the whole job, so that a batch sharded over several handles draws the worlds of the unsharded one -- mapped like numpy's
// uniform(low, high) = low + (high - low) * random_sample().  Shared by reset_kernel and the
// fused reset of the rollout kernels so both produce the same worlds.  `attempt` > 0 gives the
// redraws of a placement that is repeated until it is free (bits 48.. of the counter's high word).
template <class RS>
__host__ __device__ inline V2<double> reset_draw(const RS& rs, unsigned long long b, int e, bool is_agent, int attempt = 0) {
    uint32_t r[4];
    Philox::block(rs.seed, rs.env_base + b, rs.episode * 256ull + (unsigned long long)e + ((unsigned long long)attempt << 48), r);
    const double lim = is_agent ? 1.0 : rs.landmark_lim;
    V2<double> d;
    d.x = -lim + (lim + lim) * Philox::u01(r[0], r[1]);
    d.y = -lim + (lim + lim) * Philox::u01(r[2], r[3]);
    return d;
}

// k-th landmark index that reset_world draws with np.random.choice(world.landmarks) (the goal of
// simple_adversary.py:42 / simple_push.py:41 / simple_speaker_listener.py:40; goal_b of both agents,
// simple_reference.py:33-35; goal and key, simple_crypto.py:61-63) of world `b` in episode
// rs.episode: uniform over n_landmarks, from word k (< 4) of the Philox block with counter
// (b, episode * 256 + 255), which no entity uses.
template <class RS>
__host__ __device__ inline int reset_aux(const RS& rs, unsigned long long b, int k, int n_landmarks) {
    uint32_t r[4];
    Philox::block(rs.seed, rs.env_base + b, rs.episode * 256ull + 255ull, r);
    return (int)(((uint64_t)r[k] * (uint64_t)n_landmarks) >> 32);
}

// Random 32-bit word of world `b` in episode rs.episode (word `w` of the block the landmark indices
// come from); word 0 seeds the per-world respawn stream of scenarios whose post_step draws positions.
template <class RS>
__host__ __device__ inline int reset_stream_key(const RS& rs, unsigned long long b, int w = 0) {
    uint32_t r[4];
    Philox::block(rs.seed, rs.env_base + b, rs.episode * 256ull + 255ull, r);
    return (int)r[w];
}

// Out-of-line generator block for the rarely taken draws of scenarios that respawn landmarks: a kernel
// that tests several landmarks per step would otherwise carry one inlined copy of the ten rounds each.
// (the block comes back in registers: a reference parameter of an out-of-line function is a scratch-memory slot,
// and reading it back waits for every older memory operation of the wave)
struct PhiloxWords { uint32_t w0, w1, w2, w3; };
__device__ __noinline__ inline PhiloxWords philox_block_call(uint64_t key, uint64_t ctr_lo, uint64_t ctr_hi) {
    uint32_t out[4];
    Philox::block(key, ctr_lo, ctr_hi, out);
    return PhiloxWords{out[0], out[1], out[2], out[3]};
}

// n-th respawn position of world `b` (its index in the whole job), U(-lim, lim)^2, from the world's own stream key.  *spare
// receives the low 5 bits of the block's first word, which the position does not use.
__device__ inline V2<double> respawn_draw(int key, unsigned long long b, int n, double lim, int* spare = nullptr) {
    const PhiloxWords r = philox_block_call(0x5eed5eed00000000ull | (uint32_t)key, b, (unsigned long long)(uint32_t)n);
    if (spare) *spare = (int)(r.w0 & 31u);
    V2<double> d;
    d.x = -lim + (lim + lim) * Philox::u01(r.w0, r.w1);
    d.y = -lim + (lim + lim) * Philox::u01(r.w2, r.w3);
    return d;
}

// ---------------------------------------------------------------------------------------------
// Agent.u_noise / Agent.c_noise (multiagent/core.py:165,211): standard-normal draws scaled by the agent's
// noise level, added to the action force and to the comm state.  The reference takes them from numpy's
// global generator; here they come either from an array the caller supplies (the reference's own draws,
// for parity) or from the counter-based generator: pair k of world b at step number n is Box-Muller on the
// Philox block with key seed ^ "noise" and counter (b, n * 256 + k).
// ---------------------------------------------------------------------------------------------

// Out of line (one copy per kernel, results in registers): the float64 log / cos / sin are long.
__device__ __noinline__ inline V2<double> normal_pair_call(unsigned long long seed, unsigned long long b, unsigned long long ctr) {
    uint32_t r[4];
    Philox::block(0x6E6F697365000000ull ^ seed, b, ctr, r);
    const double u1 = 1.0 - Philox::u01(r[0], r[1]);        // (0, 1]
    const double u2 = Philox::u01(r[2], r[3]);
    const double rad = sqrt(-2.0 * log(u1));
    V2<double> z;
    z.x = rad * cos(2.0 * 3.141592653589793 * u2);
    z.y = rad * sin(2.0 * 3.141592653589793 * u2);
    return z;
}

// Pair k of world b at step t of the launch (values 2k and 2k + 1 of the row).
template <typename T>
__device__ __forceinline__ V2<T> noise_pair(const NoiseArgs& nz, long long t, long long B, long long b, int k) {
    V2<T> z;
    if (nz.draws != nullptr) {
        const T* d = static_cast<const T*>(nz.draws) + ((t * B + b) * (long long)nz.stride + 2 * k);
        z.x = d[0]; z.y = d[1];
    } else {
        const V2<double> n = normal_pair_call(nz.seed, nz.env_base + (unsigned long long)b, (nz.step0 + (unsigned long long)t) * 256ull + (unsigned long long)k);
        z.x = (T)n.x; z.y = (T)n.y;
    }
    return z;
}

}  // namespace mpe
