// snk_api.hip -- C ABI (include/snk.h) over the HIP kernels of snk_device.hpp.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC snk_api.hip -o libsnk.so
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <mutex>
#include <vector>

#include "../../include/snk.h"
#include "snk_device.hpp"

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
    g_err = msg;
    return 1;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e__));                   \
    } while (0)

}  // namespace

struct snk_handle {
    snk_params P;
    snk::HostModel H;
    snk::DevModel D;
    int n_envs, device, n, rec;
    bool v2 = true;               // register-resident solve (16 links without an obstacle); else streamed rows
    snk::DevModel* d_model = nullptr;
    float* d_recs = nullptr;
    float* d_mu = nullptr;
    // scratch for the host-buffer forms
    float* d_act = nullptr;
    float* d_obs = nullptr;
    float* d_rew = nullptr;
    uint8_t* d_done = nullptr;
    int32_t* d_sub = nullptr;
    uint8_t* d_mask = nullptr;
    float* d_tgt = nullptr;
    int32_t* d_info = nullptr;
    float* d_h = nullptr;
    float* d_linkpos = nullptr;   // allocated on first snk_link_positions
    float* d_mf = nullptr;        // contact_model 1: the persistent contact manifolds, [n_envs][2n][kMfFloats]
    float* d_rows = nullptr;      // 32-link chains: constraint rows streamed from global memory (snk_device.hpp: pgs_v1)
    unsigned long long* d_ovf = nullptr;   // contacts the solves had no room for (snk_contact_overflow): 3 counters
    float* d_box = nullptr;       // obstacle 2: the free box of every env, [n_envs][kBoxFloats] (state 13, count, manifold 24)
    int32_t* d_order = nullptr;
    bool plan = true;
    // in-launch scheduler of env_step_sched_kernel (snk_device.hpp): rings, counters, the host-mapped alarm word
    snk::Sched sched = {};
    int32_t* h_alarm = nullptr;
    int grid_waves = 0;           // resident waves of the step kernel
    int model_slot = -1;          // index into snk::g_models (constant memory); -1: none free, unscheduled kernel
    bool use_sched = true;
    size_t lds_bytes = 0;
    // timing: pool of event pairs, one pair per snk_step launch, read back in one go
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
};

namespace {

// packed_stride > 0 (snk_step_packed): obs rows of that stride, reward and done flag behind each row's observation
template <int N, bool V2>
int launch_step(snk_handle* h, float* act, float* obs, float* rew, uint8_t* done, int32_t* sub, int vec_mode,
                hipStream_t st, int packed_stride = 0) {
    const int stride = packed_stride > 0 ? packed_stride : h->D.obs_dim;
    const int packed = packed_stride > 0 ? 1 : 0;
    if (h->use_sched) {
        hipLaunchKernelGGL((snk::plan_sched_kernel<N>), dim3(1), dim3(1024), 0, st, h->d_model, h->d_recs, act, h->sched,
                           h->n_envs);
        snk::StepArgs a;
        a.recs = h->d_recs; a.mu_plane = h->d_mu; a.actions = act; a.obs = obs; a.rew = rew; a.done = done; a.substeps = sub;
        a.rows_all = h->d_rows; a.mf_all = h->d_mf; a.ovf = h->d_ovf; a.box_all = h->d_box; a.sc = h->sched;
        a.model_slot = h->model_slot; a.vec_mode = vec_mode; a.n_envs = h->n_envs;
        a.obs_stride = stride; a.packed = packed; a.pad_ = 0;
        hipLaunchKernelGGL((snk::env_step_sched_kernel<N, V2>), dim3(h->grid_waves), dim3(64), h->lds_bytes, st, a);
        return 0;
    }
    if (h->plan)
        hipLaunchKernelGGL((snk::plan_kernel<N>), dim3(1), dim3(1024), 0, st, h->d_model, h->d_recs, act, h->d_order,
                           h->n_envs);
    hipLaunchKernelGGL((snk::env_step_kernel<N, V2>), dim3(h->grid_waves), dim3(64), h->lds_bytes, st, h->d_model, h->d_recs,
                       h->d_mu, act, obs, rew, done, sub, vec_mode, h->n_envs, h->plan ? h->d_order : nullptr, h->d_rows, h->d_mf, h->d_ovf, h->d_box,
                       stride, packed);
    return 0;
}
template <int N, bool V2>
int launch_substep(snk_handle* h, const float* tgt, int k, int32_t* info, hipStream_t st) {
    hipLaunchKernelGGL((snk::substep_kernel<N, V2>), dim3(h->grid_waves), dim3(64), h->lds_bytes, st, h->d_model, h->d_recs,
                       h->d_mu, tgt, k, info, h->n_envs, h->d_rows, h->d_mf, h->d_ovf, h->d_box);
    return 0;
}
template <int N, bool V2>
int launch_reset(snk_handle* h, const uint8_t* mask, float* obs, int hard, hipStream_t st) {
    hipLaunchKernelGGL((snk::reset_kernel<N, V2>), dim3(h->n_envs), dim3(64), h->lds_bytes, st, h->d_recs, mask, obs, hard,
                       h->n_envs);
    return 0;
}
template <int N, bool V2>
int launch_obs(snk_handle* h, float* obs, float* height, hipStream_t st, float* linkpos) {
    hipLaunchKernelGGL((snk::obs_kernel<N, V2>), dim3(h->n_envs), dim3(64), h->lds_bytes, st, h->d_model, h->d_recs, obs,
                       height, linkpos, h->n_envs);
    return 0;
}
// Waves of the step kernel a CU holds: __launch_bounds__(64, 2) = 2 per SIMD, 4 SIMDs, and the CU's LDS (160 KB on
// gfx950; hipOccupancyMaxActiveBlocksPerMultiprocessor assumes 64 KB and under-counts).  Too many is harmless (late
// blocks find the queue drained), too few idles the chip.
int resident_waves(size_t bytes, int device, int* out) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    size_t lds = prop.maxSharedMemoryPerMultiProcessor;
    if (lds < 160 * 1024) lds = 160 * 1024;
    int per_cu = (int)(lds / (bytes ? bytes : 1));
    if (per_cu > 4 * SNK_LB) per_cu = 4 * SNK_LB;        // (the step kernels' launch bounds: SNK_LB waves per SIMD)
    if (const char* w = getenv("SNK_WAVES_PER_CU")) {       // occupancy experiments only (fewer resident waves than fit)
        const int v = atoi(w);
        if (v >= 1 && v < per_cu) per_cu = v;
    }
    if (per_cu < 1) per_cu = 1;
    *out = per_cu * prop.multiProcessorCount;
    return 0;
}
template <int N, bool V2>
int set_lds_attr(size_t bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&snk::env_step_kernel<N, V2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&snk::env_step_sched_kernel<N, V2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&snk::substep_kernel<N, V2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&snk::reset_kernel<N, V2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&snk::obs_kernel<N, V2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// Kernels are instantiated for the chain lengths the BASELINE configs use.
// (chain length, solve) pairs the kernels are instantiated for: 16 links with the register-resident solve, 16 and 32
// links with the streamed-row solve (32: always; 16: when the world holds an obstacle, whose contacts need it)
#define SNK_DISPATCH(h, FN, ...)                                                                            \
    ((h)->n == 16 ? ((h)->v2 ? FN<16, true>(__VA_ARGS__) : FN<16, false>(__VA_ARGS__))                      \
                  : ((h)->n == 32 ? FN<32, false>(__VA_ARGS__) : fail("unsupported n_modules (16 or 32)")))

// slots of snk::g_models, shared by the handles of this process
std::mutex g_slot_mutex;
bool g_slot_used[snk::kModelSlots] = {};
int take_model_slot() {
    std::lock_guard<std::mutex> lk(g_slot_mutex);
    for (int i = 0; i < snk::kModelSlots; i++)
        if (!g_slot_used[i]) { g_slot_used[i] = true; return i; }
    return -1;
}
void release_model_slot(int i) {
    if (i < 0) return;
    std::lock_guard<std::mutex> lk(g_slot_mutex);
    g_slot_used[i] = false;
}

// the step kernel's bounded waits (snk_device.hpp: sched_pop) raise this word when one runs out.  From then on the
// handle is poisoned: an env-step was abandoned half-way, so records, contact caches and queue no longer describe a
// simulator state.  Every call that launches on, reads or writes that state refuses; snk_destroy, snk_last_error, the
// dimension getters, the counters (snk_contact_overflow / _histogram) and the timing calls still work.
int check_alarm(const snk_handle* h) {
    if (h->h_alarm && *(volatile int32_t*)h->h_alarm)
        return fail("env-step scheduler: a bounded wait ran out (state of this handle is no longer valid; snk_destroy it)");
    return 0;
}
// the state accessors: the device idle (a step enqueued on a non-blocking stream may still be running), then the alarm
#define SNK_SYNC_ALIVE(h)                           \
    do {                                            \
        HIP_TRY(hipSetDevice((h)->device));         \
        HIP_TRY(hipDeviceSynchronize());            \
        if (check_alarm(h)) return 1;               \
    } while (0)

int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(std::string("kernel launch: ") + hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

const char* snk_last_error(void) { return g_err.c_str(); }

void snk_default_params(snk_params* p) {
    memset(p, 0, sizeof(*p));
    p->struct_size = (uint32_t)sizeof(snk_params);
    p->abi_version = SNK_ABI_VERSION;
    p->n_modules = 16;
    p->inertia_from_file = 0;
    p->default_mass = 1.0;
    p->collision_margin = 0.001;
    p->hull_sides = 32;         // PyBullet's import of a URDF <cylinder> [U]; snake.py:93 passes no URDF_USE_IMPLICIT_CYLINDER
    p->contact_model = 1;       // Bullet's persistent manifold [U] (0 + hull_sides 0: the round-1 model)
    p->self_collision = 1;      // the reference loads the snake with URDF_USE_SELF_COLLISION (snake.py:93)
    p->obstacle = 0;            // snake.py:94 has add_obstacle commented out; snake_gait_test.py:51 loads it
    p->obstacle_pos[0] = 2.0; p->obstacle_pos[1] = 0.0; p->obstacle_pos[2] = 0.1;
    p->obstacle_half[0] = 0.1; p->obstacle_half[1] = 0.4; p->obstacle_half[2] = 0.1;      // snake/block.urdf:16
    p->mu_obstacle = 0.5;       // [U]
    p->obstacle_mass = 200.0;   // snake/block.urdf:6
    p->dt = 1.0 / 240.0;
    p->gravity_z = -9.8;
    p->lin_damping = 0.04;
    p->ang_damping = 0.04;
    p->joint_damping = 0.1;
    p->max_coord_vel = 100.0;
    p->kp = 0.1;
    p->kd = 1.0;
    p->max_motor_impulse = std::numeric_limits<double>::infinity();
    p->joint_lo = -1.57;
    p->joint_hi = 1.57;
    p->limit_erp = 0.2;
    p->limit_max_impulse = 100.0;
    p->mu_link = 2.0;
    p->aniso[0] = 1.0; p->aniso[1] = 0.1; p->aniso[2] = 0.01;
    p->contact_erp = 0.08;
    p->linear_slop = 1e-5;
    p->breaking_threshold = 0.02;
    p->relative_breaking_threshold = 1;     // btCollisionDispatcher's default flags [U]
    p->cone_friction = 1;
    p->n_iterations = 50;
    p->residual_threshold = 1e-7;
    p->warm_start = 0;          // disabled in btMultiBodyConstraintSolver [U]
    p->warmstarting_factor = 0.85;
    p->friction_directions = 2;
    p->scaling_factor = 3.14159265358979323846 / 6.0;
    p->gait = 1;
    p->servo_tol = 0.05;
    p->max_counter = 40;
    p->height_threshold = 0.1;
    p->energy_dt = 1.0 / 100.0;
    p->alpha = 1.0; p->beta = 0.01; p->gamma = 0.1;
    p->term_angle = 0.5;
    p->term_index = 9;
    p->collision_force = 10.0;
    p->collision_penalty = -10.0;
    p->done_penalty = -5.0;
    p->contact_order = 0;
    p->reserved0 = 0;
}

int snk_destroy(snk_handle* h);
namespace {
// everything snk_create allocates; on failure the caller destroys the half-built handle
int init_handle(snk_handle* h, const snk_params* p, int32_t n_envs, int32_t device) {
    h->P = *p;
    h->n_envs = n_envs;
    h->device = device;
    h->n = p->n_modules;
    snk::build_host_model(*p, h->H);
    snk::build_dev_model(*p, h->H, h->D);
    // SNK_POISON=1 (tests): the kernels' LDS images and every fresh device allocation start as NaNs instead of as
    // whatever was there, so that a read of something never written shows in the outputs
    const bool poison = getenv("SNK_POISON") != nullptr;
    h->D.poison = poison ? 1 : 0;
    h->D.hist = 0;
    h->rec = h->D.rec_floats;
    // 16 links: the register-resident solve (the obstacle's contacts take slots out of the ground's 64).
    // SNK_FORCE_STREAMED=1 (diagnostics, tests): the streamed-row kernels for a 16-link handle too
    h->v2 = h->n == 16 && getenv("SNK_FORCE_STREAMED") == nullptr && p->obstacle != 2;      // (a free box: six more
                                                                                            //  components, streamed rows)
    h->lds_bytes = h->n == 16 ? (h->v2 ? sizeof(snk::Lds<16, true>) : sizeof(snk::Lds<16, false>)) : sizeof(snk::Lds<32, false>);
    int rc = SNK_DISPATCH(h, set_lds_attr, h->lds_bytes);
    if (rc) return rc;
    const size_t ne = (size_t)n_envs;
    HIP_TRY(hipMalloc(&h->d_model, sizeof(snk::DevModel)));
    HIP_TRY(hipMemcpy(h->d_model, &h->D, sizeof(snk::DevModel), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->d_recs, ne * h->rec * sizeof(float)));
    HIP_TRY(hipMemset(h->d_recs, 0, ne * h->rec * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_mu, ne * sizeof(float)));
    std::vector<float> ones(ne, 1.0f);   // plane.urdf lateral_friction = 1 [U]
    HIP_TRY(hipMemcpy(h->d_mu, ones.data(), ne * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&h->d_act, ne * h->D.act_dim * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_obs, ne * h->D.obs_dim * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_rew, ne * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_done, ne));
    HIP_TRY(hipMalloc(&h->d_sub, ne * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&h->d_mask, ne));
    HIP_TRY(hipMalloc(&h->d_tgt, ne * h->n * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_info, ne * 2 * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&h->d_h, ne * sizeof(float)));
    HIP_TRY(hipMalloc(&h->d_order, ne * sizeof(int32_t)));
    if (poison) {
        const size_t na = ne * h->D.act_dim * sizeof(float), no = ne * h->D.obs_dim * sizeof(float);
        HIP_TRY(hipMemset(h->d_act, 0xFF, na)); HIP_TRY(hipMemset(h->d_obs, 0xFF, no));
        HIP_TRY(hipMemset(h->d_rew, 0xFF, ne * sizeof(float))); HIP_TRY(hipMemset(h->d_tgt, 0xFF, ne * h->n * sizeof(float)));
        HIP_TRY(hipMemset(h->d_h, 0xFF, ne * sizeof(float)));
    }
    {
        int waves = 0;
        rc = resident_waves(h->lds_bytes, device, &waves);
        if (rc) return rc;
        h->grid_waves = waves > 0 && waves < n_envs ? waves : n_envs;
    }
    {
        // (16-link handles on the register-resident solve need them too: a substep whose contacts outgrow its 64 slots
        //  goes through the streamed-row solve in place, snk_device.hpp: substep())
        // one block of streamed constraint rows per RESIDENT WAVE (every step / substep kernel is launched with that many
        // workgroups and strides over the environments): 2048 x 112 KB = 230 MB for 32 links, whatever n_envs is
        const size_t rf = h->n == 32 ? snk::Lds<32, false>::kRowFloats : snk::Lds<16, false>::kRowFloats;
        const size_t nb = (size_t)h->grid_waves;
        const size_t bytes = nb * rf * sizeof(float);
        HIP_TRY(hipMalloc(&h->d_rows, bytes));
        HIP_TRY(hipMemset(h->d_rows, poison ? 0xFF : 0, bytes));
        // what the kernels rely on being zero for good: the last three rows of every block (the refill of a skipped
        // friction pair) and the pad columns of the M^-1 block behind the rows
        const size_t z0 = h->n == 32 ? (size_t)(snk::Lds<32, false>::kRows - 3) * snk::Lds<32, false>::kRS
                                     : (size_t)(snk::Lds<16, false>::kRows - 3) * snk::Lds<16, false>::kRS;
        const size_t zn = 3 * (size_t)(h->n == 32 ? snk::Lds<32, false>::kRS : snk::Lds<16, false>::kRS);
        const size_t m0 = h->n == 32 ? snk::Lds<32, false>::kMmOff : snk::Lds<16, false>::kMmOff;
        const size_t m1 = h->n == 32 ? snk::Lds<32, false>::kYOff : snk::Lds<16, false>::kYOff;
        if (poison)
            for (size_t e = 0; e < nb; e++) {
                HIP_TRY(hipMemsetAsync(h->d_rows + e * rf + z0, 0, zn * sizeof(float), nullptr));
                HIP_TRY(hipMemsetAsync(h->d_rows + e * rf + m0, 0, (m1 - m0) * sizeof(float), nullptr));
            }
    }
    if (p->obstacle == 2) {
        // the free box where loadURDF puts it (snake.py:84, snake_gait_test.py:51): at rest, identity orientation,
        // empty manifold
        std::vector<float> b(ne * snk::kBoxFloats, 0.f);
        for (size_t e = 0; e < ne; e++) {
            float* x = &b[e * snk::kBoxFloats];
            for (int i = 0; i < 3; i++) x[i] = (float)p->obstacle_pos[i];
            x[6] = 1.0f;
        }
        HIP_TRY(hipMalloc(&h->d_box, b.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->d_box, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&h->d_ovf, (snk::kOvfCounters + snk::kHistBins) * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(h->d_ovf, 0, (snk::kOvfCounters + snk::kHistBins) * sizeof(unsigned long long)));
    if (p->contact_model == 1) {
        const size_t bytes = ne * 2 * h->n * snk::kMfFloats * sizeof(float);
        HIP_TRY(hipMalloc(&h->d_mf, bytes));
        HIP_TRY(hipMemset(h->d_mf, 0, bytes));       // hard reset: empty manifolds
    }
    h->plan = getenv("SNK_NO_PLAN") == nullptr;
    {
        // scheduler state (snk_device.hpp: Sched)
        const char* q = getenv("SNK_QUANTUM");       // substeps per slice; 0 = whole env-steps in launch order
        int quantum = q ? atoi(q) : 1;
        h->model_slot = quantum > 0 ? take_model_slot() : -1;
        h->use_sched = h->model_slot >= 0;
        if (h->use_sched)
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(snk::g_models), &h->D, sizeof(snk::DevModel),
                                      (size_t)h->model_slot * sizeof(snk::DevModel), hipMemcpyHostToDevice));
        snk::Sched& sc = h->sched;
        sc.cap = 2u;                                  // power of two >= 2 n_envs: slot = ticket & (cap - 1) survives the
        while (sc.cap < 2u * (uint32_t)n_envs) sc.cap <<= 1;      // wrap-around of the 32-bit tickets
        sc.quantum = quantum;
        {
            const char* hy = getenv("SNK_HYST");      // experiments: see snk_device.hpp Sched::hyst
            sc.hyst = hy ? atoi(hy) : 3;
            if (sc.hyst < 1) sc.hyst = 1;
        }
        HIP_TRY(hipMalloc(&sc.head, sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&sc.tail, sizeof(uint32_t)));
        HIP_TRY(hipMemset(sc.head, 0, sizeof(uint32_t)));
        HIP_TRY(hipMemset(sc.tail, 0, sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&sc.waiting, snk::kBuckets * sizeof(int32_t)));
        HIP_TRY(hipMemset(sc.waiting, 0, snk::kBuckets * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&sc.ent, (size_t)sc.cap * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(sc.ent, 0xFF, (size_t)sc.cap * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc(&sc.counter, ne * sizeof(int32_t)));
        HIP_TRY(hipMalloc(&sc.finished, sizeof(int32_t)));
        HIP_TRY(hipMemset(sc.finished, 0, sizeof(int32_t)));
        HIP_TRY(hipHostMalloc(&h->h_alarm, 64 * sizeof(int32_t), hipHostMallocMapped));
        memset(h->h_alarm, 0, 64 * sizeof(int32_t));
        HIP_TRY(hipHostGetDevicePointer((void**)&sc.alarm, h->h_alarm, 0));
#ifdef SNK_SCHED_DEBUG
        HIP_TRY(hipMalloc(&sc.wstat, (size_t)h->grid_waves * 8 * sizeof(long long)));
#endif
    }
    // hard reset (snake.py:88-95)
    SNK_DISPATCH(h, launch_reset, h, nullptr, nullptr, 1, nullptr);
    if (check_launch()) return 1;
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}
}  // namespace

int snk_create(const snk_params* p, int32_t n_envs, int32_t device, snk_handle** out) {
    if (!p || !out) return fail("snk_create: null argument");
    if (n_envs <= 0) return fail("snk_create: n_envs must be positive");
    if (n_envs >= (1 << 24)) return fail("snk_create: n_envs must be below 2^24 (the step queue packs the env index into 24 bits)");
    if (p->struct_size != (uint32_t)sizeof(snk_params) || p->abi_version != SNK_ABI_VERSION) {
        static char msg[256];
        snprintf(msg, sizeof(msg), "snk_create: snk_params layout mismatch: the caller's struct is %u bytes, ABI version %u; this "
                 "library's is %zu bytes, version %d (fill the struct with snk_default_params of THIS library's header)",
                 p->struct_size, p->abi_version, sizeof(snk_params), SNK_ABI_VERSION);
        return fail(msg);
    }
    if (p->n_modules != 16 && p->n_modules != 32) return fail("snk_create: n_modules must be 16 or 32");
    if (p->hull_sides != 0 && (p->hull_sides < 3 || p->hull_sides > 32))
        return fail("snk_create: hull_sides must be 0 (implicit cylinder) or 3 .. 32");
    if (p->contact_model != 0 && p->contact_model != 1) return fail("snk_create: contact_model must be 0 or 1");
    if (p->obstacle < 0 || p->obstacle > 2) return fail("snk_create: obstacle must be 0, 1 (static box) or 2 (free box)");
    if (p->obstacle == 2 && p->n_modules != 16)
        return fail("snk_create: obstacle 2 (the free box) is built for n_modules 16 (its six velocity components sit behind "
                    "the snake's 22 in the 40-lane solve)");
    if (p->obstacle == 2 && !(p->obstacle_mass > 0.0)) return fail("snk_create: obstacle_mass must be positive");
    if (p->warm_start != 0 && p->warm_start != 1) return fail("snk_create: warm_start must be 0 or 1");
    if (p->friction_directions != 1 && p->friction_directions != 2) return fail("snk_create: friction_directions must be 1 or 2");
    if (p->contact_order < 0) return fail("snk_create: contact_order must be 0 (link order), 1 (reversed), 2 (Bullet's quickSort on equal keys) or k >= 3 (a fixed permutation)");
    if (p->contact_order != 0 && p->contact_model != 1)
        return fail("snk_create: contact_order needs contact_model 1 (it orders the persistent ground manifolds)");
    if (p->warm_start && p->contact_model != 1)
        return fail("snk_create: warm_start needs contact_model 1 (the impulses live in the persistent contact cache)");
    if (!(p->breaking_threshold > 0.0)) return fail("snk_create: breaking_threshold must be positive");
    if (p->term_index < 0 || p->term_index >= 3 * p->n_modules + 8) return fail("snk_create: term_index out of range");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("snk_create: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    snk_handle* h = new snk_handle();
    const int rc = init_handle(h, p, n_envs, device);
    if (rc) {
        const std::string why = snk_last_error();      // snk_destroy's own calls must not overwrite the reason
        snk_destroy(h);
        return fail(why);
    }
    *out = h;
    return 0;
}

int snk_destroy(snk_handle* h) {
    if (!h) return 0;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    void* bufs[] = {h->d_model, h->d_recs, h->d_mu, h->d_act, h->d_obs, h->d_rew, h->d_done,
                    h->d_sub, h->d_mask, h->d_tgt, h->d_info, h->d_h, h->d_order, h->d_rows, h->d_linkpos, h->d_mf, h->d_ovf, h->d_box,

                    h->sched.head, h->sched.tail, h->sched.ent, h->sched.waiting, h->sched.counter, h->sched.finished};
    for (void* b : bufs) (void)hipFree(b);
    if (h->h_alarm) (void)hipHostFree(h->h_alarm);
    release_model_slot(h->model_slot);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    delete h;
    return 0;
}

#ifdef SNK_SCHED_DEBUG
// per resident wave of the last step: ticks (100 MHz) waiting in the queue, ticks alive, slices, substeps
int snk_sched_stats(snk_handle* h, long long* out, int32_t* n_waves) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->sched.wstat, (size_t)h->grid_waves * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    *n_waves = h->grid_waves;
    return 0;
}
#endif
int32_t snk_num_envs(const snk_handle* h) { return h->n_envs; }
int32_t snk_obs_dim(const snk_handle* h) { return h->D.obs_dim; }
int32_t snk_act_dim(const snk_handle* h) { return h->D.act_dim; }
int32_t snk_state_dim(const snk_handle* h) { return h->D.state_dim; }
int32_t snk_record_floats(const snk_handle* h) { return h->rec; }

int snk_reset(snk_handle* h, const uint8_t* mask_dev, float* obs_dev, void* stream) {
    if (!h) return fail("snk_reset: null handle");
    if (check_alarm(h)) return 1;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    SNK_DISPATCH(h, launch_reset, h, mask_dev, obs_dev, 0, st);
    return check_launch();
}

int snk_step(snk_handle* h, float* actions_dev, float* obs_dev, float* rew_dev, uint8_t* done_dev,
             int32_t* substeps_dev, int32_t vec_mode, void* stream) {
    if (!h) return fail("snk_step: null handle");
    if (!actions_dev || !obs_dev || !rew_dev || !done_dev) return fail("snk_step: null buffer");
    if (check_alarm(h)) return 1;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const bool timed = 2 * h->ev_used + 1 < (int)h->ev.size();
    if (timed) HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used], st));
    SNK_DISPATCH(h, launch_step, h, actions_dev, obs_dev, rew_dev, done_dev, substeps_dev, vec_mode, st);
    if (timed) {
        HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used + 1], st));
        h->ev_used++;
    }
    return check_launch();
}

int snk_step_packed(snk_handle* h, float* actions_dev, float* packed_dev, int32_t row_stride, int32_t* substeps_dev,
                    int32_t vec_mode, void* stream) {
    if (!h) return fail("snk_step_packed: null handle");
    if (!actions_dev || !packed_dev) return fail("snk_step_packed: null buffer");
    if (row_stride < h->D.obs_dim + 2) return fail("snk_step_packed: row_stride must be at least obs_dim + 2");
    if (check_alarm(h)) return 1;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const bool timed = 2 * h->ev_used + 1 < (int)h->ev.size();
    if (timed) HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used], st));
    SNK_DISPATCH(h, launch_step, h, actions_dev, packed_dev, nullptr, nullptr, substeps_dev, vec_mode, st, row_stride);
    if (timed) {
        HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used + 1], st));
        h->ev_used++;
    }
    return check_launch();
}

int snk_timing_enable(snk_handle* h, int32_t capacity) {
    if (!h) return fail("null handle");
    HIP_TRY(hipSetDevice(h->device));
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear();
    h->ev_used = 0;
    for (int i = 0; i < 2 * capacity; i++) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    return 0;
}
int snk_timing_read(snk_handle* h, double* mean_ms, int32_t* count) {
    if (!h) return fail("null handle");
    HIP_TRY(hipSetDevice(h->device));
    double sum = 0;
    for (int i = 0; i < h->ev_used; i++) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(h->ev[2 * i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]));
        sum += ms;
    }
    if (mean_ms) *mean_ms = h->ev_used ? sum / h->ev_used : 0.0;
    if (count) *count = h->ev_used;
    h->ev_used = 0;
    return 0;
}

int snk_reset_host(snk_handle* h, const uint8_t* mask, float* obs) {
    if (!h) return fail("snk_reset_host: null handle");
    HIP_TRY(hipSetDevice(h->device));
    const size_t ne = (size_t)h->n_envs;
    if (mask) HIP_TRY(hipMemcpy(h->d_mask, mask, ne, hipMemcpyHostToDevice));
    if (obs) HIP_TRY(hipMemcpy(h->d_obs, obs, ne * h->D.obs_dim * sizeof(float), hipMemcpyHostToDevice));
    int rc = snk_reset(h, mask ? h->d_mask : nullptr, obs ? h->d_obs : nullptr, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (obs) HIP_TRY(hipMemcpy(obs, h->d_obs, ne * h->D.obs_dim * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int snk_step_host(snk_handle* h, float* actions, float* obs, float* rew, uint8_t* done, int32_t* substeps,
                  int32_t vec_mode) {
    if (!h) return fail("snk_step_host: null handle");
    if (!actions || !obs || !rew || !done) return fail("snk_step_host: null buffer");
    HIP_TRY(hipSetDevice(h->device));
    const size_t ne = (size_t)h->n_envs;
    HIP_TRY(hipMemcpy(h->d_act, actions, ne * h->D.act_dim * sizeof(float), hipMemcpyHostToDevice));
    int rc = snk_step(h, h->d_act, h->d_obs, h->d_rew, h->d_done, h->d_sub, vec_mode, nullptr);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if (check_alarm(h)) return 1;
    HIP_TRY(hipMemcpy(actions, h->d_act, ne * h->D.act_dim * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(obs, h->d_obs, ne * h->D.obs_dim * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(rew, h->d_rew, ne * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(done, h->d_done, ne, hipMemcpyDeviceToHost));
    if (substeps) HIP_TRY(hipMemcpy(substeps, h->d_sub, ne * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int snk_substep_host(snk_handle* h, const float* targets, int32_t k, int32_t* info) {
    if (!h || !targets) return fail("snk_substep_host: null argument");
    if (k < 0) return fail("snk_substep_host: k < 0");
    if (check_alarm(h)) return 1;
    HIP_TRY(hipSetDevice(h->device));
    const size_t ne = (size_t)h->n_envs;
    HIP_TRY(hipMemcpy(h->d_tgt, targets, ne * h->n * sizeof(float), hipMemcpyHostToDevice));
    SNK_DISPATCH(h, launch_substep, h, h->d_tgt, k, h->d_info, nullptr);
    if (check_launch()) return 1;
    HIP_TRY(hipDeviceSynchronize());
    if (info) HIP_TRY(hipMemcpy(info, h->d_info, ne * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

int snk_get_state(snk_handle* h, float* state, float* aux) {
    if (!h) return fail("snk_get_state: null handle");
    SNK_SYNC_ALIVE(h);
    const size_t ne = (size_t)h->n_envs;
    std::vector<float> recs(ne * h->rec);
    HIP_TRY(hipMemcpy(recs.data(), h->d_recs, recs.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int sd = h->D.state_dim, n = h->n;
    for (size_t e = 0; e < ne; e++) {
        if (state) memcpy(state + e * sd, &recs[e * h->rec], sd * sizeof(float));
        if (aux) memcpy(aux + e * (n + 2), &recs[e * h->rec + sd], (n + 2) * sizeof(float));
    }
    return 0;
}

int snk_set_state(snk_handle* h, const float* state, const float* aux) {
    if (!h) return fail("snk_set_state: null handle");
    SNK_SYNC_ALIVE(h);
    const size_t ne = (size_t)h->n_envs;
    std::vector<float> recs(ne * h->rec);
    HIP_TRY(hipMemcpy(recs.data(), h->d_recs, recs.size() * sizeof(float), hipMemcpyDeviceToHost));
    const int sd = h->D.state_dim, n = h->n;
    for (size_t e = 0; e < ne; e++) {
        if (state) memcpy(&recs[e * h->rec], state + e * sd, sd * sizeof(float));
        if (aux) memcpy(&recs[e * h->rec + sd], aux + e * (n + 2), (n + 2) * sizeof(float));
    }
    HIP_TRY(hipMemcpy(h->d_recs, recs.data(), recs.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int32_t snk_manifold_floats(const snk_handle* h) { return (h && h->d_mf) ? 2 * h->n * 29 : 0; }

// host layout per cylinder (same as the oracle's): [count, 4 x (point on the link in link coordinates 3, point on the
// ground 3, applied normal impulse)] = 29 floats; device layout: [count, 3 pad, 4 x (a3, b.x, b.y, lambda)] = kMfFloats
// (the ground point's z is the plane's, 0)
int snk_get_manifold(snk_handle* h, float* out) {
    if (!h || !out) return fail("snk_get_manifold: null argument");
    if (!h->d_mf) return fail("snk_get_manifold: this handle has contact_model 0 (no contact cache)");
    SNK_SYNC_ALIVE(h);
    const size_t ncyl = (size_t)h->n_envs * 2 * h->n;
    std::vector<float> dev(ncyl * snk::kMfFloats);
    HIP_TRY(hipMemcpy(dev.data(), h->d_mf, dev.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t c = 0; c < ncyl; c++) {
        const float* d = &dev[snk::kMfFloats * c];
        float* o = out + 29 * c;
        const int n = d[0] < 0.f ? 0 : (d[0] > 4.f ? 4 : (int)d[0]);
        o[0] = (float)n;
        for (int j = 0; j < 4; j++) {
            const bool on = j < n;       // slots past the count hold stale points: reported as zeros
            for (int r = 0; r < 3; r++) o[1 + 7 * j + r] = on ? d[4 + 6 * j + r] : 0.f;
            o[4 + 7 * j] = on ? d[7 + 6 * j] : 0.f;
            o[5 + 7 * j] = on ? d[8 + 6 * j] : 0.f;
            o[6 + 7 * j] = 0.f;
            o[7 + 7 * j] = on ? d[9 + 6 * j] : 0.f;
        }
    }
    return 0;
}
int snk_set_manifold(snk_handle* h, const float* in) {
    if (!h || !in) return fail("snk_set_manifold: null argument");
    if (!h->d_mf) return fail("snk_set_manifold: this handle has contact_model 0 (no contact cache)");
    SNK_SYNC_ALIVE(h);
    const size_t ncyl = (size_t)h->n_envs * 2 * h->n;
    std::vector<float> dev(ncyl * snk::kMfFloats, 0.f);
    for (size_t c = 0; c < ncyl; c++) {
        float* d = &dev[snk::kMfFloats * c];
        const float* o = in + 29 * c;
        d[0] = o[0] >= 0.f ? (o[0] <= 4.f ? floorf(o[0]) : 4.f) : 0.f;      // a manifold holds 0 .. 4 points, whatever the caller says
        for (int j = 0; j < 4; j++) {
            for (int r = 0; r < 3; r++) d[4 + 6 * j + r] = o[1 + 7 * j + r];
            d[7 + 6 * j] = o[4 + 7 * j];
            d[8 + 6 * j] = o[5 + 7 * j];
            d[9 + 6 * j] = o[7 + 7 * j];
        }
    }
    HIP_TRY(hipMemcpy(h->d_mf, dev.data(), dev.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int snk_get_box(snk_handle* h, float* state, float* manifold) {
    if (!h) return fail("snk_get_box: null handle");
    if (!h->d_box) return fail("snk_get_box: this handle has no free box (obstacle != 2)");
    SNK_SYNC_ALIVE(h);
    std::vector<float> b((size_t)h->n_envs * snk::kBoxFloats);
    HIP_TRY(hipMemcpy(b.data(), h->d_box, b.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < (size_t)h->n_envs; e++) {
        const float* x = &b[e * snk::kBoxFloats];
        if (state) memcpy(state + 13 * e, x, 13 * sizeof(float));
        if (manifold) {
            float* o = manifold + 29 * e;
            const int n = x[13] < 0.f ? 0 : (x[13] > 4.f ? 4 : (int)x[13]);
            o[0] = (float)n;
            for (int j = 0; j < 4; j++) {
                const bool on = j < n;
                for (int r = 0; r < 3; r++) o[1 + 7 * j + r] = on ? x[14 + 6 * j + r] : 0.f;
                o[4 + 7 * j] = on ? x[17 + 6 * j] : 0.f;
                o[5 + 7 * j] = on ? x[18 + 6 * j] : 0.f;
                o[6 + 7 * j] = 0.f;
                o[7 + 7 * j] = on ? x[19 + 6 * j] : 0.f;
            }
        }
    }
    return 0;
}
int snk_set_box(snk_handle* h, const float* state, const float* manifold) {
    if (!h) return fail("snk_set_box: null handle");
    if (!h->d_box) return fail("snk_set_box: this handle has no free box (obstacle != 2)");
    SNK_SYNC_ALIVE(h);
    std::vector<float> b((size_t)h->n_envs * snk::kBoxFloats);
    HIP_TRY(hipMemcpy(b.data(), h->d_box, b.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < (size_t)h->n_envs; e++) {
        float* x = &b[e * snk::kBoxFloats];
        if (state) memcpy(x, state + 13 * e, 13 * sizeof(float));
        if (manifold) {
            const float* o = manifold + 29 * e;
            x[13] = o[0] >= 0.f ? (o[0] <= 4.f ? floorf(o[0]) : 4.f) : 0.f;      // 0 .. 4 cached points
            for (int j = 0; j < 4; j++) {
                for (int r = 0; r < 3; r++) x[14 + 6 * j + r] = o[1 + 7 * j + r];
                x[17 + 6 * j] = o[4 + 7 * j];
                x[18 + 6 * j] = o[5 + 7 * j];
                x[19 + 6 * j] = o[7 + 7 * j];
            }
        }
    }
    HIP_TRY(hipMemcpy(h->d_box, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int snk_contact_overflow(snk_handle* h, uint64_t* out) {
    if (!h || !out) return fail("snk_contact_overflow: null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v[3];
    HIP_TRY(hipMemcpy(v, h->d_ovf, sizeof(v), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) out[i] = (uint64_t)v[i];
    return 0;
}

int32_t snk_contact_histogram_bins(void) { return snk::kHistBins; }
int snk_contact_histogram_enable(snk_handle* h, int32_t on) {
    if (!h) return fail("snk_contact_histogram_enable: null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    h->D.hist = on ? 1 : 0;
    HIP_TRY(hipMemcpy(h->d_model, &h->D, sizeof(snk::DevModel), hipMemcpyHostToDevice));
    if (h->model_slot >= 0)
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(snk::g_models), &h->D, sizeof(snk::DevModel),
                                  (size_t)h->model_slot * sizeof(snk::DevModel), hipMemcpyHostToDevice));
    return 0;
}
int snk_contact_histogram(snk_handle* h, uint64_t* out, int32_t reset) {
    if (!h || !out) return fail("snk_contact_histogram: null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> v(snk::kHistBins);
    HIP_TRY(hipMemcpy(v.data(), h->d_ovf + snk::kOvfCounters, v.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < snk::kHistBins; i++) out[i] = (uint64_t)v[i];
    if (reset) HIP_TRY(hipMemset(h->d_ovf + snk::kOvfCounters, 0, v.size() * sizeof(unsigned long long)));
    return 0;
}

int snk_get_obs(snk_handle* h, float* obs) {
    if (!h || !obs) return fail("snk_get_obs: null argument");
    SNK_SYNC_ALIVE(h);
    SNK_DISPATCH(h, launch_obs, h, h->d_obs, nullptr, nullptr, nullptr);
    if (check_launch()) return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(obs, h->d_obs, (size_t)h->n_envs * h->D.obs_dim * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int snk_mean_height(snk_handle* h, float* out) {
    if (!h || !out) return fail("snk_mean_height: null argument");
    SNK_SYNC_ALIVE(h);
    SNK_DISPATCH(h, launch_obs, h, nullptr, h->d_h, nullptr, nullptr);
    if (check_launch()) return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_h, (size_t)h->n_envs * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int snk_link_positions(snk_handle* h, float* out) {
    if (!h || !out) return fail("snk_link_positions: null argument");
    SNK_SYNC_ALIVE(h);
    const size_t bytes = (size_t)h->n_envs * 3 * (h->n + 1) * sizeof(float);
    if (!h->d_linkpos) HIP_TRY(hipMalloc(&h->d_linkpos, bytes));
    SNK_DISPATCH(h, launch_obs, h, nullptr, nullptr, nullptr, h->d_linkpos);
    if (check_launch()) return 1;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->d_linkpos, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int snk_set_ground_friction(snk_handle* h, const float* mu) {
    if (!h || !mu) return fail("snk_set_ground_friction: null argument");
    SNK_SYNC_ALIVE(h);
    HIP_TRY(hipMemcpy(h->d_mu, mu, (size_t)h->n_envs * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

int snk_joint3_reaction_fz(snk_handle* h, float* out) {
    if (!h || !out) return fail("snk_joint3_reaction_fz: null argument");
    SNK_SYNC_ALIVE(h);
    const size_t ne = (size_t)h->n_envs;
    std::vector<float> recs(ne * h->rec);
    HIP_TRY(hipMemcpy(recs.data(), h->d_recs, recs.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < ne; e++) out[e] = recs[e * h->rec + 15 + 3 * h->n];      // behind prev_x in the record
    return 0;
}

int snk_get_ground_friction(snk_handle* h, float* mu) {
    if (!h || !mu) return fail("snk_get_ground_friction: null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(mu, h->d_mu, (size_t)h->n_envs * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int snk_debug_set_tickets(snk_handle* h, uint32_t base) {
    if (!h) return fail("snk_debug_set_tickets: null handle");
    SNK_SYNC_ALIVE(h);
    HIP_TRY(hipMemcpy(h->sched.head, &base, sizeof(base), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->sched.tail, &base, sizeof(base), hipMemcpyHostToDevice));
    return 0;
}

int snk_debug_raise_alarm(snk_handle* h) {
    if (!h) return fail("snk_debug_raise_alarm: null handle");
    if (!h->h_alarm) return fail("snk_debug_raise_alarm: this handle has no alarm word");
    // the word is host-mapped memory: the host stores what a wave whose bounded wait ran out would (sched_alarm);
    // a step still running sees it at its next bounded wait and drains
    __atomic_store_n(h->h_alarm, 1, __ATOMIC_SEQ_CST);
    return 0;
}

int snk_selftest(int32_t device) {
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("snk_selftest: no such HIP device");
    HIP_TRY(hipSetDevice(device));
    float* d = nullptr;
    HIP_TRY(hipMalloc(&d, 136 * sizeof(float)));
    hipLaunchKernelGGL(snk::selftest_kernel, dim3(1), dim3(64), 0, nullptr, d);
    if (check_launch()) return 1;
    float o[136];
    HIP_TRY(hipMemcpy(o, d, sizeof(o), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    if (o[4] != 32.f * 33.f / 2.f || o[5] != (64.f * 65.f - 32.f * 33.f) / 2.f) {
        char buf[120];
        snprintf(buf, sizeof(buf), "selftest mismatch: half_reduce lower=%g upper=%g", o[4], o[5]);
        return fail(buf);
    }
    for (int l = 0; l < 64; l++) {
        // half_swap(x, y): a = [x.lo, y.lo], b = [x.hi, y.hi]
        float wa = l < 32 ? (float)(l + 1) : 100.f * (float)(l - 32 + 1);
        float wb = l < 32 ? (float)(l + 32 + 1) : 100.f * (float)(l + 1);
        if (o[8 + l] != wa || o[72 + l] != wb) {
            char buf[160];
            snprintf(buf, sizeof(buf), "selftest mismatch: half_swap lane %d a=%g (want %g) b=%g (want %g)", l, o[8 + l], wa,
                     o[72 + l], wb);
            return fail(buf);
        }
    }
    const float s22 = 22.f * 23.f / 2.f, s64 = 64.f * 65.f / 2.f, s38 = 38.f * 39.f / 4.f;
    if (o[0] != s22 || o[1] != s64 || o[2] != 18.f || o[3] != s38) {
        char buf[160];
        snprintf(buf, sizeof(buf), "selftest mismatch: sum32=%g (want %g) sum64=%g (want %g) bcast=%g (want 18) sum38=%g (want %g)",
                 o[0], s22, o[1], s64, o[2], o[3], s38);
        return fail(buf);
    }
    return 0;
}

int snk_params_derived(const snk_params* p, double* out) {
    if (!p || !out) return fail("snk_params_derived: null argument");
    if (p->n_modules != 16 && p->n_modules != 32) return fail("snk_params_derived: n_modules must be 16 or 32");
    snk::HostModel H;
    snk::DevModel D;
    snk::build_host_model(*p, H);
    snk::build_dev_model(*p, H, D);
    out[0] = D.break_thr; out[1] = D.cyl_r; out[2] = D.cyl_hl; out[3] = D.cyl_zoff; out[4] = D.margin; out[5] = D.obs_thr;
    return 0;
}

int snk_model_describe(const snk_handle* h, double* out_bodies, double* out_origins) {
    if (!h) return fail("snk_model_describe: null handle");
    const int n = h->n;
    double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, op[3] = {0, 0, 0};
    for (int b = 0; b <= n; b++) {
        if (out_bodies) {
            double* o = out_bodies + 10 * b;
            o[0] = h->H.mass[b];
            for (int i = 0; i < 3; i++) o[1 + i] = h->H.com[b][i];
            const double* I = h->H.Ib[b];
            o[4] = I[0]; o[5] = I[1]; o[6] = I[2]; o[7] = I[4]; o[8] = I[5]; o[9] = I[8];
        }
        if (b >= 1) {
            double t[3], Rn[9];
            snk::detail::mv(Rp, h->H.pfix[b], t);
            for (int i = 0; i < 3; i++) op[i] += t[i];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    double s = 0;
                    for (int k = 0; k < 3; k++) s += Rp[3 * i + k] * h->H.Rfix[b][3 * k + j];
                    Rn[3 * i + j] = s;
                }
            memcpy(Rp, Rn, sizeof(Rn));
        }
        if (out_origins)
            for (int i = 0; i < 3; i++) out_origins[3 * b + i] = op[i];
    }
    return 0;
}

}  // extern "C"
