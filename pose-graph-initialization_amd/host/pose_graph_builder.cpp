// pose_graph_builder.cpp -- implementation of the C++ host layer (links against libpgi.so).
// HIP is used here for device buffers and copies only; every computation is a C-ABI call.
#include "distributed.hpp"
#include "graph_traversal.hpp"
#include "reconstruction.hpp"
#include "tracklets.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

#include <pthread.h>
#include <sched.h>

namespace reconstruction {

namespace {
struct DevBuf {
    void* p = nullptr;
    explicit DevBuf(size_t bytes) {
        if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) throw PgiError("hipMalloc failed");
    }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};
// Device memory of one processFeatures wave: blocks handed out by bumping an offset and recycled at the start of the next
// wave (two dozen hipMalloc / hipFree pairs per wave cost more than most of the wave's kernels).
struct DevPool {
    std::vector<std::pair<void*, size_t>> blocks;
    size_t used = 0;  // within the last block
    ~DevPool() { for (auto& b : blocks) (void)hipFree(b.first); }
    void* take(size_t bytes) {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        if (blocks.empty() || used + bytes > blocks.back().second) {
            const size_t nb = std::max(bytes, blocks.empty() ? (size_t)16 << 20 : 2 * blocks.back().second);
            void* q = nullptr;
            if (hipMalloc(&q, nb) != hipSuccess) throw PgiError("hipMalloc failed");
            blocks.emplace_back(q, nb);
            used = 0;
        }
        void* r = static_cast<char*>(blocks.back().first) + used;
        used += bytes;
        return r;
    }
    void reset() {  // the device must be idle on these blocks
        if (blocks.size() > 1) {  // one block of the combined size from now on
            size_t total = 0;
            for (auto& b : blocks) { total += b.second; (void)hipFree(b.first); }
            blocks.clear();
            void* q = nullptr;
            if (hipMalloc(&q, total) != hipSuccess) throw PgiError("hipMalloc failed");
            blocks.emplace_back(q, total);
        }
        used = 0;
    }
};
struct WaveBuf {  // DevBuf's interface over pool memory
    void* p;
    WaveBuf(DevPool& pool, size_t bytes) : p(pool.take(bytes)) {}
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};
// fn(i) for i in [0, n) on up to `threads` host threads (contiguous, equally sized index ranges; the caller's thread takes the
// first).  What the reference does with `#pragma omp parallel for num_threads(kCoreNumber)` (pose_graph_builder.h:391-392).
template <class Fn>
void parallelFor(size_t n, size_t threads, Fn fn) {
    const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(threads, n), std::max(1u, std::thread::hardware_concurrency())));
    auto work = [&](size_t t) {
        const size_t a = n * t / nt, b = n * (t + 1) / nt;
        for (size_t i = a; i < b; ++i) fn(i);
    };
    if (nt == 1) { work(0); return; }
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (size_t t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread& th : pool) th.join();
}
void h2d(void* d, const void* h, size_t n) {
    if (n && hipMemcpy(d, h, n, hipMemcpyHostToDevice) != hipSuccess) throw PgiError("hipMemcpy H2D failed");
}
void d2h(void* h, const void* d, size_t n) {
    if (n && hipMemcpy(h, d, n, hipMemcpyDeviceToHost) != hipSuccess) throw PgiError("hipMemcpy D2H failed");
}
}  // namespace

namespace {
// f64 AoS rows of many pairs -> flattened f32 SoA batch on the device
struct DeviceBatch {
    std::vector<uint64_t> off;
    size_t rows = 0;
    std::unique_ptr<DevBuf> x1, y1, x2, y2, doff, thr, guess, has, edges, masks;
    pgi_batch b{};
};
}  // namespace

uint32_t PoseGraphBuilder::worldSize() const { return hostComm ? hostComm->world() : 1u; }
uint32_t PoseGraphBuilder::worldRank() const { return hostComm ? hostComm->rank() : 0u; }

// The reference's guess path refits every accepted chained pose on the rows inside the UN-squared getInliers bound
// (graph_traversal.h:149,164 -- at 0.4 px and f = 1000 that is ~24 px), so a pose with a handful of true inliers yields an
// edge.  Faithful by default; when such edges are more than 5 % of the accepted guesses the run says so once (stderr,
// rank 0; PGI_QUIET silences it) and counts it ("[Pose estimation] Quirk-only warning").
void PoseGraphBuilder::warnQuirkOnlyGuesses(size_t quirkOnly, size_t acceptedGuesses) {
    if (!acceptedGuesses || quirkOnly * 20 <= acceptedGuesses) return;
    statistics.addCount("[Pose estimation] Quirk-only warning", 1, 1);
    if (worldRank() != 0 || std::getenv("PGI_QUIET")) return;
    std::fprintf(stderr,
                 "pose-graph builder: %zu of %zu accepted A* pose guesses (%.0f %%) have fewer than kMinimumInlierNumber rows inside the "
                 "squared bound (1.5 thr)^2 -- these edges exist only through the reference's un-squared getInliers comparison "
                 "(graph_traversal.h:164) and are usually wrong; PoseGraphBuilder::setRotationGuidedGuesses(true) re-estimates chained "
                 "poses instead (see DESIGN.md section 4-4)\n",
                 quirkOnly, acceptedGuesses, 100.0 * (double)quirkOnly / (double)acceptedGuesses);
}

// A few host threads that stay alive between calls (the reference's kCoreNumber OpenMP team, pose_graph_builder.h:391-392):
// run(n, fn) calls fn(i) for i in [0, n) in contiguous, equally sized index ranges, the caller taking the first.
// The CPUs of the NUMA node a HIP device hangs on (sysfs; empty when that cannot be told)
static std::vector<int> deviceLocalCpus(int dev, int* node_out) {
    std::vector<int> cpus;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) { (void)hipGetLastError(); return cpus; }
    for (char* c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
    int node = -1;
    if (std::FILE* f = std::fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node_out) *node_out = node;
    if (node < 0) return cpus;
    char path[96];
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    std::FILE* f = std::fopen(path, "r");
    if (!f) return cpus;
    int a = 0, b = 0;
    while (std::fscanf(f, "%d", &a) == 1) {  // "0-63,128-191"
        b = a;
        int ch = std::fgetc(f);
        if (ch == '-') {
            if (std::fscanf(f, "%d", &b) != 1) break;
            ch = std::fgetc(f);
        }
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) cpus.push_back(c);
        if (ch != ',') break;
    }
    std::fclose(f);
    return cpus;
}

// The stream the context enqueues on NOW (pgi_set_stream may have moved it off the default stream): uploads on the staging
// copy stream are ordered against it with events, never against a stream assumed to be the default one.
static int engineDevice(pgi_ctx* ctx) {
    int d = -1;
    Engine::check(pgi_get_device(ctx, &d));
    return d;
}
static hipStream_t engineStream(pgi_ctx* ctx) {
    void* s = nullptr;
    Engine::check(pgi_get_stream(ctx, &s));
    return (hipStream_t)s;
}

// What `numactl --cpunodebind` does for a one-process-per-GPU launch: the calling thread (and every thread it starts later:
// the host team, the runtime's helpers) may only run on the CPUs of the NUMA node the device hangs on, and the memory it
// touches first lands there.  The builder's host side reads the caller's matrices, writes the page-locked ring and drives
// the copy engine: on a two-socket host, a process living on the far socket ran config 5 of the dense V = 5000 scene in
// 0.16-0.18 s against 0.14 s (scripts/numa_probe.sh; binding only the team's workers changed nothing -- the data and the
// calling thread are what matters).  Call it FIRST, before the inputs are read.  Returns the node, -1 when nothing was done
// (single node, sysfs unreadable, refused by a cpuset, PGI_HOST_NUMA=0).
int PoseGraphBuilder::bindProcessToDeviceNode(int device) {
    if (const char* e = std::getenv("PGI_HOST_NUMA")) if (e[0] == '0') return -1;
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    int node = -1;
    const std::vector<int> cpus = deviceLocalCpus(dev, &node);
    if (cpus.empty()) return -1;
    // An explicit launcher binding wins (taskset, torchrun/SLURM per-rank pinning, a cpuset): only CPUs the process may
    // already run on are kept.  Nothing is done when the inherited mask is already inside the node, or does not meet it.
    cpu_set_t inherited, set;
    CPU_ZERO(&inherited);
    if (sched_getaffinity(0, sizeof inherited, &inherited) != 0) return -1;
    CPU_ZERO(&set);
    for (int c : cpus) if (CPU_ISSET(c, &inherited)) CPU_SET(c, &set);
    if (CPU_COUNT(&set) == 0) return -1;                           // bound elsewhere on purpose: leave it
    if (CPU_COUNT(&set) == CPU_COUNT(&inherited)) return node;     // already a subset of the node
    return sched_setaffinity(0, sizeof set, &set) == 0 ? node : -1;
}

class HostPool {
   public:
    explicit HostPool(size_t threads) {
        const size_t n = std::max<size_t>(1, std::min<size_t>(threads, std::max(1u, std::thread::hardware_concurrency())));
        for (size_t t = 1; t < n; ++t) workers.emplace_back([this, t] { loop(t); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> l(mu);
            quit = true;
        }
        wake.notify_all();
        for (std::thread& th : workers) th.join();
    }
    size_t size() const { return workers.size() + 1; }
    template <class Fn>
    void run(size_t n, Fn fn) {
        const size_t nt = std::min(size(), std::max<size_t>(n, 1));
        std::function<void(size_t)> body = [&](size_t t) {
            if (t >= nt) return;
            const size_t a = n * t / nt, b = n * (t + 1) / nt;
            for (size_t i = a; i < b; ++i) fn(i);
        };
        if (nt == 1) { body(0); return; }
        std::lock_guard<std::mutex> one(callers);  // one job at a time: the pool is shared by every builder on the device
        {
            std::lock_guard<std::mutex> l(mu);
            job = &body;
            pending = workers.size();
            ++generation;
        }
        wake.notify_all();
        struct Join {  // also when fn throws on the calling thread: the workers still run `body`, which lives in this frame
            HostPool& p;
            ~Join() {
                std::unique_lock<std::mutex> l(p.mu);
                p.done.wait(l, [&] { return p.pending == 0; });
                p.job = nullptr;
            }
        } join{*this};
        body(0);
    }

   private:
    void loop(size_t t) {
        uint64_t seen = 0;
        for (;;) {
            std::function<void(size_t)>* mine;
            {
                std::unique_lock<std::mutex> l(mu);
                wake.wait(l, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
                mine = job;
            }
            (*mine)(t);
            {
                std::lock_guard<std::mutex> l(mu);
                if (--pending == 0) done.notify_one();
            }
        }
    }
    std::vector<std::thread> workers;
    std::mutex mu, callers;
    std::condition_variable wake, done;
    std::function<void(size_t)>* job = nullptr;
    size_t pending = 0;
    uint64_t generation = 0;
    bool quit = false;
};

// Staging of estimatePoses (round 4: a pipeline instead of one block).  The reference hands over one cv::Mat N x 4 CV_64F
// per pair (pageable, 32 B per row); the device wants the flattened f32 SoA.  At SURVEY 8d's size -- 10^5 pairs, 7.7 * 10^7
// rows, 2.45 GB of doubles -- converting everything into one page-locked block first cost 0.22 s of a 0.41 s run (a third of
// it the 1.5 GB page-locked allocation itself).  Now the block's pairs are cut into chunks of ~4 M rows; the host team
// converts chunk c + 1 into one of three page-locked ring buffers while chunk c travels on a copy stream and the kernels of
// chunk c - 1 run; ring, device block, streams and events are grow-only and shared by every builder of the process.
struct PoseGraphBuilder::Staging {
    static constexpr int kRing = 3;
    // TWO device blocks and two page-locked mirrors of the per-pair arrays (round 5): while the kernels of one wave work on
    // block `active`, the rows of the scheduler's NEXT wave are converted and uploaded into the other one (prefetch below)
    void* devBlk[2] = {nullptr, nullptr};
    size_t devBytes[2] = {0, 0};
    void* smallBlk[2] = {nullptr, nullptr};  // page-locked per-pair arrays (offsets, thresholds, guesses, screening models)
    size_t smallBytes[2] = {0, 0};
    int active = 0;
    struct Prefetched {   // the rows of `pairs[0..P)` -- this rank's block of them -- lie in block `which`; preDone marks their arrival
        const void* first = nullptr;
        size_t P = 0, rows = 0;
        int which = 0;
        bool valid = false;
    } pre;
    hipEvent_t preDone = nullptr;
    void* ring[kRing] = {nullptr, nullptr, nullptr};
    size_t ring_bytes = 0;  // each
    hipStream_t copy = nullptr;
    hipEvent_t up[kRing] = {nullptr, nullptr, nullptr};
    hipEvent_t smallUp = nullptr;
    std::unique_ptr<HostPool> pool;
    std::unique_ptr<HostPool> commitPool;  // the team that writes a wave's edges into the pose graph (its own: the scheduler's helper
                                           // thread converts the NEXT wave's rows on `pool` at that very time)
    std::mutex busy;  // one estimatePoses at a time per process
    // processFeatures (round 5): the feature arena (every view's keypoints, descriptors and prepared copies: 3.5 GB at config
    // 3's size) and the waves' device pool are LENT to one run at a time and never released -- a fresh hipMalloc of the arena
    // per call took 0.3-0.45 s in one repetition out of four on the driver's box (0.003 s in the others), which was the whole
    // of the "slow repetitions" of VERDICT r4 (bench.py: all_repetitions_stage_s).  A second builder running at the same time
    // finds the lock taken and allocates privately, as before.
    std::mutex featBusy;
    void* featArena = nullptr;
    size_t featArenaBytes = 0;
    DevPool featPool;
    char* lendArena(size_t bytes) {  // featBusy held by the caller
        if (bytes > featArenaBytes) {
            if (featArena) (void)hipFree(featArena);
            featArena = nullptr; featArenaBytes = 0;
            if (hipMalloc(&featArena, bytes + bytes / 8) != hipSuccess) throw PgiError("hipMalloc failed");
            featArenaBytes = bytes + bytes / 8;
        }
        return (char*)featArena;
    }
    ~Staging() {
        if (featArena) (void)hipFree(featArena);
        for (void* d : devBlk) if (d) (void)hipFree(d);
        for (void* h : smallBlk) if (h) (void)hipHostFree(h);
        if (preDone) (void)hipEventDestroy(preDone);
        for (void* r : ring) if (r) (void)hipHostFree(r);
        for (hipEvent_t e : up) if (e) (void)hipEventDestroy(e);
        if (smallUp) (void)hipEventDestroy(smallUp);
        if (copy) (void)hipStreamDestroy(copy);
    }
    void init(size_t threads) {
        if (!copy) {  // highest priority: a copy that shares a priority with queued K1 workgroups is starved by them
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (hipStreamCreateWithPriority(&copy, hipStreamNonBlocking, greatest) != hipSuccess) throw PgiError("hipStreamCreate failed");
        }
        for (hipEvent_t& e : up)
            if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) throw PgiError("hipEventCreate failed");
        if (!smallUp && hipEventCreateWithFlags(&smallUp, hipEventDisableTiming) != hipSuccess) throw PgiError("hipEventCreate failed");
        if (!preDone && hipEventCreateWithFlags(&preDone, hipEventDisableTiming) != hipSuccess) throw PgiError("hipEventCreate failed");
        if (!pool || pool->size() < std::min<size_t>(threads, std::max(1u, std::thread::hardware_concurrency()))) pool.reset(new HostPool(threads));
        if (!commitPool) commitPool.reset(new HostPool(std::min<size_t>(8, std::max<size_t>(1, threads))));
    }
    static void growHost(void*& p, size_t& have, size_t want) {
        if (want <= have) return;
        if (p) (void)hipHostFree(p);
        p = nullptr; have = 0;
        if (hipHostMalloc(&p, want + want / 4, hipHostMallocDefault) != hipSuccess) throw PgiError("hipHostMalloc failed");
        have = want + want / 4;
    }
    void reserve(int which, size_t smallWant, size_t ringBytes, size_t devWant) {
        growHost(smallBlk[which], smallBytes[which], smallWant);
        if (ringBytes > ring_bytes) {
            size_t have = 0;
            for (void*& r : ring) { have = ring_bytes; growHost(r, have, ringBytes); }
            ring_bytes = have;
        }
        if (devWant > devBytes[which]) {
            if (devBlk[which]) (void)hipFree(devBlk[which]);
            devBlk[which] = nullptr; devBytes[which] = 0;
            if (hipMalloc(&devBlk[which], devWant + devWant / 4) != hipSuccess) throw PgiError("hipMalloc failed");
            devBytes[which] = devWant + devWant / 4;
        }
    }
    // Process-wide and never released: page-locking a few hundred MB costs tens of milliseconds, and a process that builds
    // one pose graph after another (or one builder per configuration) should pay that once.  (Freed by the process's exit;
    // an explicit teardown would have to run before the HIP runtime's own, which static destruction order does not promise.)
    // One per DEVICE: the block, the events and the copy stream belong to the device that was current when they were made
    // (pgi_create(device) allows several devices per process); the builder's engine names the device.
    static std::shared_ptr<Staging> shared(int device) {
        static std::mutex m;
        static std::map<int, std::shared_ptr<Staging>>* inst = new std::map<int, std::shared_ptr<Staging>>();
        std::lock_guard<std::mutex> l(m);
        std::shared_ptr<Staging>& s = (*inst)[device];
        if (!s) s.reset(new Staging());
        return s;
    }
};

#define HIP_OK(x)                                                                                       \
    do {                                                                                                \
        const hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) throw PgiError(std::string(#x) + ": " + hipGetErrorString(e_));           \
    } while (0)

size_t PoseGraphBuilder::estimatePoses(const std::vector<ViewPair>& pairs, PoseGraph& poseGraph_, uint64_t seed,
                                       std::vector<pgi_edge>* edges_out, bool screenGuesses, pgi_edge* d_edges_out,
                                       const std::function<void(size_t, size_t)>* prepareGuesses,
                                       std::function<size_t()>* deferredInsertion,
                                       const std::function<const std::vector<ViewPair>*()>* nextWave, bool rowsOnly) {
    const size_t P = pairs.size();
    if (deferredInsertion) *deferredInsertion = nullptr;
    if (!P) return 0;
    typedef std::chrono::steady_clock Clock;
    const Clock::time_point t0 = Clock::now();
    Clock::time_point tp = t0;
    const bool timing = std::getenv("PGI_HOST_TIMING") != nullptr;  // stderr: wall clock per phase of this function
    auto mark = [&](const char* what) {  // phase clocks: RunningStatistics "[Pose estimation] <phase>" (+ stderr on request)
        if (rowsOnly) return;  // (a prefetch runs on a helper thread: the statistics belong to the calling one)
        const Clock::time_point now = Clock::now();
        const double sec = std::chrono::duration<double>(now - tp).count();
        statistics.addTime(std::string("[Pose estimation] ") + what, sec, 1);
        if (timing) std::fprintf(stderr, "[estimatePoses] %-34s %8.3f ms\n", what, 1e3 * sec);
        tp = now;
    };
    // this rank's contiguous, row-balanced block [lo, hi) of the wave
    const uint32_t world = worldSize(), rank = worldRank();
    std::vector<uint64_t> rowsPerPair(P);
    for (size_t i = 0; i < P; ++i) rowsPerPair[i] = (uint64_t)pairs[i].correspondences.rows;
    const std::vector<std::pair<size_t, size_t>> blocks = dist::shardBounds(rowsPerPair, world);
    const size_t lo = blocks[rank].first, hi = blocks[rank].second, L = hi - lo;
    std::vector<uint64_t> off(L + 1, 0);
    for (size_t k = 0; k < L; ++k) off[k + 1] = off[k] + rowsPerPair[lo + k];
    const size_t rows = off[L];
    // chunks of whole pairs, ~chunkRows rows each (PGI_HOST_CHUNK_ROWS overrides the default for experiments)
    size_t chunkRows = (size_t)4 << 20;
    if (const char* e = std::getenv("PGI_HOST_CHUNK_ROWS")) chunkRows = std::max<size_t>(1, std::strtoull(e, nullptr, 10));
    struct Chunk { size_t k0, k1; uint32_t maxCorr; };
    std::vector<Chunk> chunks;
    size_t maxChunkRows = 0;
    for (size_t k = 0; k < L;) {
        size_t k1 = k;
        uint32_t mc = 0;
        while (k1 < L && (k1 == k || off[k1 + 1] - off[k] <= chunkRows)) {
            mc = std::max(mc, (uint32_t)rowsPerPair[lo + k1]);
            ++k1;
        }
        chunks.push_back(Chunk{k, k1, mc});
        maxChunkRows = std::max<size_t>(maxChunkRows, off[k1] - off[k]);
        k = k1;
    }
    // (a short last chunk joins its predecessor: transfers of a few kilobytes are copy kernels, and a launch pays a fixed wind-down)
    if (chunks.size() >= 2 && off[chunks.back().k1] - off[chunks.back().k0] < chunkRows / 4) {
        const Chunk tail = chunks.back();
        chunks.pop_back();
        chunks.back().k1 = tail.k1;
        chunks.back().maxCorr = std::max(chunks.back().maxCorr, tail.maxCorr);
        maxChunkRows = std::max<size_t>(maxChunkRows, off[chunks.back().k1] - off[chunks.back().k0]);
    }
    // device layout (256-byte aligned pieces): the whole block's SoA rows, the per-pair arrays, masks, screening scratch;
    // the page-locked `small` block mirrors the per-pair part [o_off, o_small_end)
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_x1 = 0, o_y1 = o_x1 + up(rows * 4), o_x2 = o_y1 + up(rows * 4), o_y2 = o_x2 + up(rows * 4),
                 o_off = o_y2 + up(rows * 4), o_thr = o_off + up((L + 1) * 8), o_guess = o_thr + up(L * 8),
                 o_has = o_guess + up(L * 96), o_Eg = o_has + up(L), o_tau = o_Eg + up(L * 72), o_small_end = o_tau + up(L * 8),
                 o_cnt = o_small_end, o_masks = o_cnt + up(L * 4), o_all = o_masks + up(rows),
                 dev_total = o_all + (d_edges_out ? 0 : up(P * sizeof(pgi_edge)));  // the gathered table, unless the caller brings one
    if (!staging) staging = Staging::shared(engineDevice(engine->get()));
    // (rowsOnly: the PREFETCH of the scheduler's next wave, called from inside the running wave's call, which holds the lock)
    std::unique_lock<std::mutex> stagingBusy(staging->busy, std::defer_lock);
    if (!rowsOnly) stagingBusy.lock();
    HIP_OK(hipSetDevice(engineDevice(engine->get())));  // this thread's allocations, stream and events belong to the engine's device
    staging->init(kCoreNumber ? kCoreNumber : 1);
    if (rowsOnly && !L) {
        // This rank's block of the NEXT wave is empty (fewer pairs than ranks, or one heavy pair): there is nothing to upload, and
        // nothing below -- the collective, the download, the insertion, the statistics -- is a prefetch's business: the peers
        // issue no second all-gather, and the other block holds no records of this wave.
        staging->pre.valid = false;
        return 0;
    }
    // Which of the two blocks: a prefetch fills the one the running wave does not use; a wave whose rows were prefetched
    // works where they lie; every other call on the active one.
    const bool prefetched = !rowsOnly && staging->pre.valid && staging->pre.first == (const void*)pairs.data() && staging->pre.P == P &&
                            staging->pre.rows == rows;
    if (!rowsOnly) {
        if (prefetched) staging->active = staging->pre.which;
        staging->pre.valid = false;  // consumed, or stale: either way nobody else may take it
    }
    const int which = rowsOnly ? 1 - staging->active : staging->active;
    staging->reserve(which, o_small_end - o_off, maxChunkRows * 16, dev_total);
    char* const hs = (char*)staging->smallBlk[which] - o_off;  // so that hs + o_* addresses the page-locked mirror
    char* const db = (char*)staging->devBlk[which];
    double *thr = (double*)(hs + o_thr), *guess = (double*)(hs + o_guess), *Eg = (double*)(hs + o_Eg), *tau2 = (double*)(hs + o_tau);
    uint8_t* has = (uint8_t*)(hs + o_has);
    if (!prefetched) {
        memcpy(hs + o_off, off.data(), (L + 1) * 8);
        for (size_t k = 0; k < L; ++k) thr[k] = pairs[lo + k].normalizedThreshold;  // every pair, also those without rows
    }
    // Pose guesses: known up front (the caller filled poseGuesses), or produced per launch group by `prepareGuesses` -- the
    // scheduler's A* searches for the pairs [first, last) of `pairs` -- right before the group's rows are converted, so
    // that the host searches for group g + 1 while the device estimates group g.  Either way the guess arrays of a group
    // are filled and uploaded with the group.
    bool may_guess = prepareGuesses != nullptr;
    for (size_t k = 0; k < L && !may_guess && !rowsOnly; ++k) may_guess = !pairs[lo + k].poseGuesses.empty();
    const bool screen = may_guess && screenGuesses;
    std::vector<uint8_t> screened(screen ? L : 0, 0);  // this block's pairs that carried a chained pose into the screening launch
    std::vector<uint32_t> guessInliers;                // their inlier counts under the SQUARED bound (1.5 thr)^2
    auto fillGuesses = [&](size_t g0, size_t g1) {     // per-pair guess arrays of the pairs [g0, g1) of the block
        for (size_t k = g0; k < g1; ++k) {
            const ViewPair& vp = pairs[lo + k];
            has[k] = 0;
            if (!vp.poseGuesses.empty()) {  // the last guess wins (pose_graph_builder.h:974-1029)
                const SE3d& g = vp.poseGuesses.back();
                for (int c2 = 0; c2 < 9; ++c2) guess[12 * k + c2] = g.R[c2];
                for (int c2 = 0; c2 < 3; ++c2) guess[12 * k + 9 + c2] = g.t[c2];
                has[k] = 1;
            } else {
                for (int c2 = 0; c2 < 12; ++c2) guess[12 * k + c2] = 0.0;
            }
            if (!screen) continue;
            // InTraversalPoseTester::test for every chained pose, one launch per group:
            // E = [t]x R (pose_utils.h:74-86), bound (1.5 thr)^2 (:798), accepted at 5 inliers (:809)
            screened[k] = has[k];
            for (int c = 0; c < 9; ++c) Eg[9 * k + c] = 0.0;
            tau2[k] = 0.0;
            if (!has[k]) { Eg[9 * k] = 1.0; continue; }
            const Matrix3d E = pose::getEssentialMatrixFromRelativePose(vp.poseGuesses.back());
            for (int c = 0; c < 9; ++c) Eg[9 * k + c] = E[c];
            tau2[k] = (1.5 * thr[k]) * (1.5 * thr[k]);
        }
    };
    // the gathered table (P records); this rank's block is written in place at [lo, hi)
    pgi_edge* const d_all = d_edges_out ? d_edges_out : (pgi_edge*)(db + o_all);  // (in the grow-only block: no hipMalloc per wave)
    mark("per-pair arrays (host)");
    double convertSeconds = 0, searchSeconds = 0;
    if (L) {
        hipStream_t copy = staging->copy;
        if (prefetched) {  // offsets, thresholds and every row are in the block already: the engine's stream waits for their arrival
            HIP_OK(hipStreamWaitEvent(engineStream(engine->get()), staging->preDone, 0));
        } else {
            HIP_OK(hipMemcpyAsync(db + o_off, hs + o_off, o_guess - o_off, hipMemcpyHostToDevice, copy));  // offsets, thresholds
            if (!rowsOnly) {
                HIP_OK(hipEventRecord(staging->smallUp, copy));
                HIP_OK(hipStreamWaitEvent(engineStream(engine->get()), staging->smallUp, 0));  // the stream the engine enqueues on
            }
        }
        float* const dcol[4] = {(float*)(db + o_x1), (float*)(db + o_y1), (float*)(db + o_x2), (float*)(db + o_y2)};
        size_t groupK0 = 0, groupChunks = 0, groupTarget = 1;
        uint32_t groupMaxCorr = 0;
        for (size_t c = 0; c < chunks.size(); ++c) {
            const Chunk& ch = chunks[c];
            if (groupChunks == 0 && may_guess && !rowsOnly) {  // a new launch group starts here: its guesses first (A* on the host team)
                const size_t last = std::min(chunks.size(), c + groupTarget) - 1;
                const size_t g0 = ch.k0, g1 = chunks[last].k1;
                const Clock::time_point ts = Clock::now();
                if (prepareGuesses) (*prepareGuesses)(lo + g0, lo + g1);
                searchSeconds += std::chrono::duration<double>(Clock::now() - ts).count();
                fillGuesses(g0, g1);
            }
            const int slot = (int)(c % Staging::kRing);
            const size_t r0 = off[ch.k0], cr = off[ch.k1] - r0;
            if (!prefetched) HIP_OK(hipEventSynchronize(staging->up[slot]));  // its previous upload (of this call or an earlier one) has left the buffer
            float* const hb = (float*)staging->ring[slot];
            const Clock::time_point tc = Clock::now();
            if (cr && !prefetched) {
                // the team converts contiguous row ranges of the chunk; a range starts inside the pair holding its first row
                const size_t parts = std::min<size_t>(staging->pool->size(), cr / 16384 + 1);
                staging->pool->run(parts, [&](size_t t) {
                    const size_t a = r0 + cr * t / parts, z = r0 + cr * (t + 1) / parts;
                    size_t k = (size_t)(std::upper_bound(off.begin() + ch.k0, off.begin() + ch.k1 + 1, a) - off.begin()) - 1;
                    for (size_t r = a; r < z; ++k) {
                        const size_t pairEnd = std::min<size_t>(off[k + 1], z);
                        const double* q = pairs[lo + k].correspondences.ptr((int)(r - off[k]));
                        float *x1 = hb + (r - r0), *y1 = x1 + cr, *x2 = y1 + cr, *y2 = x2 + cr;
                        for (size_t i = 0, m = pairEnd - r; i < m; ++i, q += 4) {
                            x1[i] = (float)q[0]; y1[i] = (float)q[1]; x2[i] = (float)q[2]; y2[i] = (float)q[3];
                        }
                        r = pairEnd;
                    }
                });
                for (int a = 0; a < 4; ++a)
                    HIP_OK(hipMemcpyAsync(dcol[a] + r0, hb + (size_t)a * cr, cr * 4, hipMemcpyHostToDevice, copy));
            }
            convertSeconds += std::chrono::duration<double>(Clock::now() - tc).count();
            if (!prefetched) HIP_OK(hipEventRecord(staging->up[slot], copy));
            if (rowsOnly) continue;  // a prefetch uploads and nothing else
            if (!prefetched) HIP_OK(hipStreamWaitEvent(engineStream(engine->get()), staging->up[slot], 0));
            // Kernels are launched per GROUP of uploaded chunks -- 1, 2, then 4 chunks: a launch pays a fixed wind-down while its
            // last workgroups finish (0.9 ms, DESIGN.md section 7 "The drain"), so the first launch comes early and the later
            // ones are large.
            groupMaxCorr = std::max(groupMaxCorr, ch.maxCorr);
            ++groupChunks;
            if (groupChunks < groupTarget && c + 1 < chunks.size()) continue;
            const size_t g0 = groupK0, g1 = ch.k1, n = g1 - g0;
            pgi_batch b{};
            b.d_x1 = dcol[0]; b.d_y1 = dcol[1]; b.d_x2 = dcol[2]; b.d_y2 = dcol[3];  // offsets are absolute rows of the block
            b.d_offsets = (const uint64_t*)(db + o_off) + g0; b.d_thr = (const double*)(db + o_thr) + g0;
            b.n_pairs = (uint32_t)n; b.max_corr = groupMaxCorr; b.pair_id_base = lo + g0; b.seed = seed;  // ids = positions in `pairs`
            if (may_guess) {  // the group's guess arrays travel now (one event: the ring slots' events are per chunk)
                HIP_OK(hipMemcpyAsync(db + o_guess + g0 * 96, hs + o_guess + g0 * 96, n * 96, hipMemcpyHostToDevice, copy));
                HIP_OK(hipMemcpyAsync(db + o_has + g0, hs + o_has + g0, n, hipMemcpyHostToDevice, copy));
                if (screen) {
                    HIP_OK(hipMemcpyAsync(db + o_Eg + g0 * 72, hs + o_Eg + g0 * 72, n * 72, hipMemcpyHostToDevice, copy));
                    HIP_OK(hipMemcpyAsync(db + o_tau + g0 * 8, hs + o_tau + g0 * 8, n * 8, hipMemcpyHostToDevice, copy));
                }
                HIP_OK(hipEventRecord(staging->smallUp, copy));
                HIP_OK(hipStreamWaitEvent(engineStream(engine->get()), staging->smallUp, 0));
                if (screen) {
                    Engine::check(pgi_score_pose_batch(engine->get(), &b, (const double*)(db + o_Eg) + 9 * g0, (const double*)(db + o_tau) + g0,
                                                       (uint32_t*)(db + o_cnt) + g0, nullptr));
                    Engine::check(pgi_screen_guesses(engine->get(), (const uint32_t*)(db + o_cnt) + g0, 5, (uint8_t*)(db + o_has) + g0, (uint32_t)n));
                }
                b.d_guess_Rt = (const double*)(db + o_guess) + 12 * g0;
                b.d_has_guess = (const uint8_t*)(db + o_has) + g0;
            }
            Engine::check(pgi_estimate_pose_batch(engine->get(), &b, d_all + lo + g0, (uint8_t*)(db + o_masks)));
            groupK0 = g1;
            groupChunks = 0;
            groupMaxCorr = 0;
            groupTarget = std::min<size_t>(groupTarget * 2, 4);
        }
        if (rowsOnly) {  // the descriptor the next wave's own call recognises its rows by
            HIP_OK(hipEventRecord(staging->preDone, copy));
            staging->pre.first = (const void*)pairs.data();
            staging->pre.P = P;
            staging->pre.rows = rows;
            staging->pre.which = which;
            staging->pre.valid = true;
            if (rowsOnlyConvertSeconds) *rowsOnlyConvertSeconds = convertSeconds;  // (RunningStatistics belongs to the calling thread)
            return 0;
        }
        mark("convert + upload + launch (chunks)");
        statistics.addTime("[Pose estimation] of which row conversion (host team)", convertSeconds, 1);
        if (prepareGuesses) statistics.addTime("[Pose estimation] of which path searches (host team)", searchSeconds, 1);
    }
    // the path's one exchange step (no-op copy in a single process)
    std::vector<uint32_t> counts(world);
    for (uint32_t r = 0; r < world; ++r) counts[r] = (uint32_t)(blocks[r].second - blocks[r].first);
    // The scheduler's NEXT wave (round 5): formed now and its rows converted and uploaded into the other block WHILE this wave's
    // kernels run -- the host used to sit in the wait below and then spent 6-7 ms per wave on exactly that before the first
    // launch of the next wave.  What needs this wave's commit -- the A* guesses -- is still computed after it.
    // The upload runs on a helper thread (it drives the host team, which nothing else needs until the next wave's A*) while THIS
    // thread waits for the kernels, gathers, downloads and inserts -- joined before this call returns (it works on the staging
    // this call has locked).
    std::thread prefetcher;
    std::exception_ptr prefetchError;
    double prefetchConvertSeconds = 0;
    struct JoinPrefetcher {
        std::thread& t;
        ~JoinPrefetcher() { if (t.joinable()) t.join(); }
    } joinPrefetcher{prefetcher};
    if (nextWave && !rowsOnly && L) {
        if (const std::vector<ViewPair>* nw = (*nextWave)()) {
            if (!nw->empty())
                prefetcher = std::thread([this, nw, &poseGraph_, &prefetchError, &prefetchConvertSeconds]() {
                    try {
                        rowsOnlyConvertSeconds = &prefetchConvertSeconds;
                        estimatePoses(*nw, poseGraph_, 0, nullptr, false, nullptr, nullptr, nullptr, nullptr, /*rowsOnly*/ true);
                    } catch (...) {
                        prefetchError = std::current_exception();
                    }
                });
        }
        mark("next wave: formation (its rows upload behind this wave's kernels, download and insertion)");
    }
    Engine::check(pgi_synchronize(engine->get()));
    mark("wait for the kernels");
    Engine::check(pgi_allgather_edges(engine->get(), d_all + lo, counts.data(), d_all));
    Engine::check(pgi_synchronize(engine->get()));
    mark("all-gather of the edge records");
    std::vector<pgi_edge> edges(P);
    d2h(edges.data(), d_all, P * sizeof(pgi_edge));
    if (screen) {
        guessInliers.resize(L);
        d2h(guessInliers.data(), db + o_cnt, L * 4);
    }
    mark("download");
    size_t added = 0, inliers = 0;
    for (size_t i = 0; i < P; ++i) inliers += edges[i].n_inl;
    // the edges in pair order (pose_graph_builder.h:645-654; a failed pair is skipped, :641-642), one locked batch
    HostPool* team = staging->commitPool.get();
    auto insertAll = [&pairs, &poseGraph_, P, team](const std::vector<pgi_edge>& e) {
        std::vector<PoseGraph::NewEdge> items;
        items.reserve(P);
        for (size_t i = 0; i < P; ++i) {
            if (e[i].status != PGI_EDGE_OK) continue;
            items.push_back(PoseGraph::NewEdge{pairs[i].src, pairs[i].dst, (double)e[i].n_inl / (double)std::max(1, pairs[i].correspondences.rows),
                                               e[i].R, e[i].t});
        }
        // (the records and adjacency lists of a wave are written by the host team: 260 -> ~80 ns per edge at 10^4 edges per wave)
        static const bool serial = [] { const char* e = std::getenv("PGI_COMMIT_TEAM"); return e && std::atoi(e) == 0; }();
        if (serial) poseGraph_.addEdges(items.data(), items.size());
        else poseGraph_.addEdges(items.data(), items.size(), team->size(),
                                 [team](size_t parts, const std::function<void(size_t)>& fn) { team->run(parts, fn); });
        return items.size();
    };
    if (deferredInsertion && edges_out) {
        // The caller runs it -- on another thread, next to the rotation averaging, which reads the DEVICE table and never the
        // host graph -- and joins before it returns; *edges_out holds the records by then (moved below) and outlives the call.
        *deferredInsertion = [insertAll, edges_out]() { return insertAll(*edges_out); };
    } else {
        added = insertAll(edges);
    }
    mark("pose graph insertion");
    if (prefetcher.joinable()) {
        prefetcher.join();
        rowsOnlyConvertSeconds = nullptr;
        statistics.addTime("[Pose estimation] of which row conversion of the NEXT wave (host team, on the helper thread)", prefetchConvertSeconds, 1);
        mark("next wave: rest of its row upload (exposed)");
        if (prefetchError) {
            staging->pre.valid = false;
            std::rethrow_exception(prefetchError);
        }
    }
    // Guard for the reference-faithful guess path (guess_quirk = 1, graph_traversal.h:149,164): a chained pose passes the
    // 5-inlier tester, then getInliers compares SQUARED residuals with the UN-squared bound and almost any pose collects
    // kMinimumInlierNumber rows.  Count the accepted guesses that would have failed under the squared bound: those edges
    // exist only because of the quirk (on thin graphs they are what spoils the averaged rotations).
    if (screen) {
        uint64_t quirkOnly = 0;
        for (size_t k = 0; k < L; ++k)
            if (screened[k] && edges[lo + k].used_guess && edges[lo + k].status == PGI_EDGE_OK && guessInliers[k] < kMinimumInlierNumber)
                ++quirkOnly;
        lastQuirkOnlyGuesses = quirkOnly;
    } else {
        lastQuirkOnlyGuesses = 0;
    }
    if (world > 1 && screenGuesses) {  // host-side bookkeeping, 8 bytes per rank
        uint64_t sum = 0;
        for (uint64_t v : hostComm->allgather(lastQuirkOnlyGuesses)) sum += v;
        lastQuirkOnlyGuesses = sum;
    }
    if (screenGuesses) statistics.addCount("[Pose estimation] Quirk-only guesses", lastQuirkOnlyGuesses, P);
    // observability keys of pose_graph_builder.h:636-638 (one timed event per batch, one run per pair)
    statistics.addTime("[Pose estimation]", std::chrono::duration<double>(Clock::now() - t0).count(), P);
    statistics.addCount("[Pose estimation] Runs", P, P);
    statistics.addCount("[Pose estimation] Inlier number", inliers, P);
    if (edges_out) *edges_out = std::move(edges);
    return added;
}

PoseGraphBuilder::GlobalRotations PoseGraphBuilder::estimateAndAverage(const std::vector<ViewPair>& pairs, PoseGraph& poseGraph_,
                                                                        size_t numViews, uint64_t seed,
                                                                        std::vector<pgi_edge>* edges_out,
                                                                        const pgi_rotavg_params* rotavgParams) {
    GlobalRotations out;
    out.rotations.assign(numViews, Matrix3d{{1, 0, 0, 0, 1, 0, 0, 0, 1}});
    const size_t P = pairs.size();
    if (!P || !numViews) return out;
    for (const ViewPair& vp : pairs) {
        poseGraph_.addVertex(vp.src);
        poseGraph_.addVertex(vp.dst);
    }
    DevBuf table(P * sizeof(pgi_edge));
    // The pose graph is filled on a second thread WHILE the device averages the rotations (10^5 edges: 0.03 s of host work
    // -- 200-byte records into fresh memory -- next to 0.02 s of averaging that reads only the device table).
    std::vector<pgi_edge> edgesLocal;
    std::vector<pgi_edge>* edgesHeld = edges_out ? edges_out : &edgesLocal;
    std::function<size_t()> insertion;
    estimatePoses(pairs, poseGraph_, seed, edgesHeld, false, table.as<pgi_edge>(), nullptr, &insertion);
    double insertSeconds = 0;
    std::thread inserter;
    if (insertion)
        inserter = std::thread([&]() {
            const std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
            insertion();
            insertSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
        });
    struct Join {
        std::thread& t;
        ~Join() { if (t.joinable()) t.join(); }
    } join{inserter};
    std::vector<uint32_t> src(P), dst(P), rows(P);
    for (size_t i = 0; i < P; ++i) {
        src[i] = (uint32_t)pairs[i].src;
        dst[i] = (uint32_t)pairs[i].dst;
        rows[i] = (uint32_t)pairs[i].correspondences.rows;
    }
    static_assert(sizeof(Matrix3d) == 72, "Matrix3d must be 9 packed doubles");
    const std::chrono::steady_clock::time_point t_avg = std::chrono::steady_clock::now();
    Engine::check(pgi_rotation_average_edges(engine->get(), table.as<pgi_edge>(), src.data(), dst.data(), rows.data(), (uint32_t)P,
                                             (uint32_t)numViews, rotavgParams, out.rotations[0].data(), &out.iterations,
                                             &out.edgesUsed));
    statistics.addTime("[Rotation averaging]", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_avg).count(), 1);
    if (inserter.joinable()) inserter.join();
    statistics.addTime("[Pose estimation] pose graph insertion (beside the averaging)", insertSeconds, 1);
    return out;
}

PoseGraphBuilder::GlobalRotations PoseGraphBuilder::averageRotations(const PoseGraph& poseGraph_, size_t numViews,
                                                                      const pgi_rotavg_params* rotavgParams) {
    GlobalRotations out;
    out.rotations.assign(numViews, Matrix3d{{1, 0, 0, 0, 1, 0, 0, 0, 1}});
    if (!numViews) return out;
    // the edge list in insertion order, gathered straight from the graph's record array by the host team (an id list plus a
    // locked look-up and a 208-byte copy per edge took 6 ms of the 15 at 10^5 edges)
    std::vector<pgi_rot_edge> re(poseGraph_.numEdges());
    {
        const size_t E = re.size(), parts = E >= 16384 ? std::min<size_t>(8, std::max<size_t>(1, kCoreNumber)) : 1;
        parallelFor(parts, parts, [&](size_t t) {
            poseGraph_.forEachEdge(E * t / parts, E * (t + 1) / parts, [&](size_t k, const PoseGraphEdge& e) {
                pgi_rot_edge r{};
                r.src = (uint32_t)e.getSourceId();
                r.dst = (uint32_t)e.getDestinationId();
                for (int c = 0; c < 9; ++c) r.R[c] = e.getValue().getRotation()[c];
                r.weight = e.getScore();
                re[k] = r;
            });
        });
    }
    out.edgesUsed = (uint32_t)re.size();
    Engine::check(pgi_rotation_average(engine->get(), re.data(), (uint32_t)re.size(), (uint32_t)numViews, rotavgParams,
                                       out.rotations[0].data(), &out.iterations));
    return out;
}

PoseGraphBuilder::RunStatistics PoseGraphBuilder::run(std::vector<ViewPair>& cand, PoseGraph& poseGraph_, size_t waveSize,
                                                      const SimilarityTable* similarityTable, uint64_t seedBase) {
    RunStatistics st;
    typedef std::chrono::steady_clock Clock;
    if (staging) staging->pre.valid = false;  // (rows a run that ended by an exception may have left uploaded belong to nobody)
    // descending similarity, ties by (src,dst): the order the reference pops its heap.  An index order is sorted, not the
    // candidate records themselves (10^5 records with their matrices' headers: a stable sort moved each a dozen times).
    const Clock::time_point tSort = Clock::now();
    // (the keys are copied out once and sorted as one flat array: a comparator that reached into the 10^5 candidate records
    //  through an index cost twice as much; the position in the caller's list is the last key, which makes plain std::sort stable)
    struct OrderKey { double similarity; ViewId src, dst; uint32_t index; };
    std::vector<OrderKey> keys(cand.size());
    for (size_t i = 0; i < keys.size(); ++i) keys[i] = OrderKey{cand[i].similarity, cand[i].src, cand[i].dst, (uint32_t)i};
    auto before = [](const OrderKey& a, const OrderKey& b) {
        if (a.similarity != b.similarity) return a.similarity > b.similarity;
        if (a.src != b.src) return a.src < b.src;
        if (a.dst != b.dst) return a.dst < b.dst;
        return a.index < b.index;
    };
    // (a total order -- the index breaks every tie -- so any sorting method gives the same array: eight segments on eight
    //  threads, then three levels of pairwise merges, also threaded; 10^5 keys: 7 ms on one thread)
    {
        const size_t n = keys.size();
        const size_t parts = n >= 32768 ? std::min<size_t>(8, std::max<size_t>(1, kCoreNumber)) : 1;
        if (parts <= 1) {
            std::sort(keys.begin(), keys.end(), before);
        } else {
            std::vector<size_t> cut(parts + 1);
            for (size_t t = 0; t <= parts; ++t) cut[t] = n * t / parts;
            parallelFor(parts, parts, [&](size_t t) { std::sort(keys.begin() + cut[t], keys.begin() + cut[t + 1], before); });
            for (size_t width = 1; width < parts; width *= 2) {
                const size_t merges = (parts + 2 * width - 1) / (2 * width);
                parallelFor(merges, merges, [&](size_t m) {
                    const size_t a = 2 * width * m, mid = std::min(a + width, parts), z = std::min(a + 2 * width, parts);
                    if (mid < z) std::inplace_merge(keys.begin() + cut[a], keys.begin() + cut[mid], keys.begin() + cut[z], before);
                });
            }
        }
    }
    // what wave formation reads of a candidate, in heap order: the formation streams through this array and never touches the
    // candidate records (10^5 cache misses on the 100-byte records were two thirds of its 13 ms; gathered here by the team)
    struct Ordered { ViewId src, dst; double similarity, normalizedThreshold; const double* rows; int count; };
    std::vector<Ordered> ordered(cand.size());
    {
        const size_t n = ordered.size(), parts = n >= 32768 ? std::min<size_t>(8, std::max<size_t>(1, kCoreNumber)) : 1;
        parallelFor(parts, parts, [&](size_t t) {
            for (size_t i = n * t / parts; i < n * (t + 1) / parts; ++i) {
                const ViewPair& vp = cand[keys[i].index];
                ordered[i] = Ordered{vp.src, vp.dst, vp.similarity, vp.normalizedThreshold, vp.correspondences.ptr(), vp.correspondences.rows};
            }
        });
    }
    std::vector<OrderKey>().swap(keys);
    statistics.addTime("[Scheduler] candidate order", std::chrono::duration<double>(Clock::now() - tSort).count(), 1);
    ViewId maxId = 0;
    for (const ViewPair& vp : cand) maxId = std::max(maxId, std::max(vp.src, vp.dst));
    VisibilityTable visibilityTable(maxId + 1);  // :366-367
    for (const EdgeId& id : poseGraph_.getEdgeIds()) visibilityTable.addLink(id.first, id.second);
    const bool pathFinding = kUsePathFinding && similarityTable != nullptr;
    const uint32_t world = worldSize(), rank = worldRank();
    std::vector<ViewPair> wave, nextWave;
    uint64_t seed = seedBase;
    // called by estimatePoses while the wave's kernels run: forms the NEXT wave (speculatively, see the loop below) so that its
    // rows can be uploaded meanwhile; assigned once formWave exists
    std::function<const std::vector<ViewPair>*()> formNextWhileEstimating = []() -> const std::vector<ViewPair>* { return nullptr; };
    auto flush = [&]() {
        if (wave.empty()) return;
        // findPath (:785-862) on the graph committed by the previous waves.  The searches of a wave are independent (the graph
        // does not change until the wave commits): they run on kCoreNumber threads, lock-free (PoseGraph::forEachNeighbourFrozen),
        // and -- group by group, inside estimatePoses -- while the device already estimates the previous group of the wave.
        // Every rank searches only for the pairs it estimates (estimatePoses asks for ranges of its own block).
        std::unique_ptr<ImageSimilarityHeuristics> heuristics;
        std::unique_ptr<AStarTraversal<ImageSimilarityHeuristics>> traversal;
        std::vector<uint32_t> touchedOf, foundOf;
        std::vector<uint8_t> searchedOf;
        double searchSeconds = 0;
        std::function<void(size_t, size_t)> searchRange;
        if (pathFinding) {
            if (!staging) staging = Staging::shared(engineDevice(engine->get()));
            {
                std::lock_guard<std::mutex> stagingBusy(staging->busy);  // init may replace the team another builder is using
                HIP_OK(hipSetDevice(engineDevice(engine->get())));
                staging->init(kCoreNumber ? kCoreNumber : 1);
            }
            heuristics.reset(new ImageSimilarityHeuristics(*similarityTable));
            traversal.reset(new AStarTraversal<ImageSimilarityHeuristics>(&poseGraph_, *heuristics, kTraversalHeuristicsWeight, 0.0, kMaximumSearchDepth));
            traversal->setGraphFrozen(true);
            touchedOf.assign(wave.size(), 0);
            foundOf.assign(wave.size(), 0);
            searchedOf.assign(wave.size(), 0);
            for (ViewPair& vp : wave) vp.poseGuesses.clear();
            searchRange = [&](size_t first, size_t last) {
                const Clock::time_point ts = Clock::now();
                staging->pool->run(last - first, [&](size_t k) {  // (the host team that also converts the rows)
                    const size_t i = first + k;
                    ViewPair& vp = wave[i];
                    if (!visibilityTable.hasLink(vp.src, vp.dst)) return;  // kAreViewsVisible (:456-457, 568)
                    std::vector<ViewId> path;
                    size_t touched = 0, found = 0;
                    bool exists = false;
                    traversal->getPath(vp.src, vp.dst, path, vp.poseGuesses, touched, found, exists);
                    searchedOf[i] = 1;
                    touchedOf[i] = (uint32_t)touched;
                    foundOf[i] = (uint32_t)found;
                });
                searchSeconds += std::chrono::duration<double>(Clock::now() - ts).count();
            };
        }
        std::vector<pgi_edge> edges;
        const size_t added = estimatePoses(wave, poseGraph_, seed++, &edges, /*screenGuesses*/ pathFinding && !rotationGuidedGuesses, nullptr,
                                           pathFinding ? &searchRange : nullptr, nullptr, &formNextWhileEstimating);
        if (pathFinding) {
            struct Tally { uint64_t searched = 0, touched = 0, found = 0; } tally;
            for (size_t i = 0; i < wave.size(); ++i) {  // per-pair tallies summed in pair order
                tally.searched += searchedOf[i];
                tally.touched += touchedOf[i];
                tally.found += foundOf[i];
            }
            if (world > 1) {  // the searches' tallies of all ranks (host-side bookkeeping, 24 bytes per rank)
                const std::vector<Tally> all = hostComm->allgather(tally);
                tally = Tally();
                for (const Tally& t : all) { tally.searched += t.searched; tally.touched += t.touched; tally.found += t.found; }
            }
            st.pathsSearched += tally.searched;
            st.touchedNodes += tally.touched;
            st.pathsFound += tally.found;
            // :599-602
            statistics.addTime("[A*]", searchSeconds, tally.searched);
            statistics.addCount("[A*] Runs", tally.searched, tally.searched);
            statistics.addCount("[A*] Touched nodes", tally.touched, tally.searched);
            statistics.addCount("[A*] Paths tested", tally.found, tally.searched);
        }
        const Clock::time_point t1 = Clock::now();
        for (size_t i = 0; i < wave.size(); ++i) {
            st.hypotheses += edges[i].iters;
            st.posesFromGuess += edges[i].used_guess;
            if (edges[i].status == PGI_EDGE_OK) visibilityTable.addLink(wave[i].src, wave[i].dst);  // :692
        }
        st.quirkOnlyGuesses += lastQuirkOnlyGuesses;
        statistics.addTime("[Visibility update]", std::chrono::duration<double>(Clock::now() - t1).count(), added);  // :698-699
        statistics.addCount("[Visibility update] Runs", added, added);
        st.edgesAdded += added;
        st.pairsProcessed += wave.size();
        ++st.waves;
        wave.clear();
    };
    // Wave formation (the next candidates' records are prefetched: the order is by similarity, the records lie in the caller's
    // order -- 0.017 -> 0.008 s at 10^5 candidates.  Forming wave w + 1 on a second thread while wave w is estimated -- possible
    // when the list names no pair twice, since then only edges the graph had BEFORE the run can exclude a candidate -- was built
    // and measured: the formation left the critical path, 0.008 -> 0.0025 s, but the runs were not faster, 0.155-0.18 s
    // against 0.144-0.16 s on the dense V = 5000 scene; removed.)
    size_t cursor = 0;
    std::vector<uint32_t> pick;          // candidates of the batch being admitted (positions in `cand`)
    std::vector<ViewId> pickSrc, pickDst;
    std::vector<uint8_t> admit;
    auto formWave = [&](std::vector<ViewPair>& wave) {
        // In batches of what the wave still lacks: the candidates that pass the list's own filters (similarity, row count) are
        // checked against the graph and get their vertices under ONE lock (PoseGraph::admitPairs); the order of the reference's
        // tests (:426-431 before :550-551) does not matter, both only skip.
        while (cursor < cand.size() && wave.size() < waveSize) {
            pick.clear(); pickSrc.clear(); pickDst.clear();
            for (; cursor < cand.size() && wave.size() + pick.size() < waveSize; ++cursor) {
                const Ordered& vp = ordered[cursor];
                if (vp.similarity < kSimilarityThreshold) { cursor = cand.size(); break; }  // heap holds sim >= threshold only
                if ((size_t)vp.count < kMinimumPointNumber) continue;   // :550-551
                pick.push_back((uint32_t)cursor);
                pickSrc.push_back(vp.src);
                pickDst.push_back(vp.dst);
            }
            admit.assign(pick.size(), 0);
            poseGraph_.admitPairs(pickSrc.data(), pickDst.data(), pick.size(), admit.data());   // :426-431 + addVertex x 2
            for (size_t k = 0; k < pick.size(); ++k) {
                if (!admit[k]) continue;
                const Ordered& vp = ordered[pick[k]];
                // the wave holds HEADERS over the candidates' matrices (cv::Mat semantics): the rows stay where the caller put them and
                // are released when the caller releases its list -- a wave that owned them paid for unmapping 2.4 GB of matrices inside
                // the run (0.09 s of a 0.25 s run at 10^5 pairs)
                ViewPair header;
                header.src = vp.src;
                header.dst = vp.dst;
                header.similarity = vp.similarity;
                header.normalizedThreshold = vp.normalizedThreshold;
                header.correspondences = CorrespondenceMatrix::viewOf(vp.rows, vp.count);
                wave.push_back(std::move(header));
            }
        }
    };
    // room for every candidate's edge up front: committing 10^5 edges wave by wave re-housed the 208-byte records a handful of
    // times (a third of the insertion's cost)
    poseGraph_.reserveEdges(cand.size());
    double formSeconds = 0, flushSeconds = 0;
    // The NEXT wave is formed while this wave's kernels run, i.e. BEFORE this wave's edges are in the graph.  Formation asks the
    // graph one thing -- "is this candidate an edge already, in either direction?" -- so the early answer is wrong only for a
    // candidate that the list names twice with the first instance in the running wave.  That is checked after the commit
    // (PoseGraph::anyEdgeBetween over the speculative wave): if it ever happens the speculative wave and its uploaded rows are
    // dropped and the wave is formed again the sequential way, so the result is the sequential one in every case.  (Vertices
    // added by the early formation are those the later formation adds: an excluded candidate's vertices exist -- it is an edge.)
    size_t cursorBeforeNext = 0;
    bool speculated = false;
    static const bool prefetchOn = [] { const char* e = std::getenv("PGI_WAVE_PREFETCH"); return !e || e[0] != '0'; }();
    formNextWhileEstimating = [&]() -> const std::vector<ViewPair>* {
        if (!prefetchOn) return nullptr;
        const Clock::time_point tf = Clock::now();
        cursorBeforeNext = cursor;
        nextWave.clear();
        formWave(nextWave);
        speculated = true;
        formSeconds += std::chrono::duration<double>(Clock::now() - tf).count();
        return &nextWave;
    };
    std::vector<ViewId> chkSrc, chkDst;
    for (;;) {
        const Clock::time_point tf = Clock::now();
        if (speculated) {
            speculated = false;
            chkSrc.resize(nextWave.size());
            chkDst.resize(nextWave.size());
            for (size_t i = 0; i < nextWave.size(); ++i) { chkSrc[i] = nextWave[i].src; chkDst[i] = nextWave[i].dst; }
            if (poseGraph_.anyEdgeBetween(chkSrc.data(), chkDst.data(), chkSrc.size())) {
                cursor = cursorBeforeNext;  // a candidate listed twice: redo the formation against the committed graph
                nextWave.clear();
                if (staging) staging->pre.valid = false;
                formWave(wave);
            } else {
                wave.swap(nextWave);    // (the vector's buffer travels: the prefetch is recognised by it)
                nextWave.clear();
            }
        } else {
            formWave(wave);
        }
        const Clock::time_point tw = Clock::now();
        formSeconds += std::chrono::duration<double>(tw - tf).count();
        if (wave.empty()) break;
        flush();  // estimates and commits `wave`, and clears it
        flushSeconds += std::chrono::duration<double>(Clock::now() - tw).count();
    }
    statistics.addTime("[Scheduler] wave formation", formSeconds, 1);
    statistics.addTime("[Scheduler] waves (search, estimate, commit)", flushSeconds, 1);
    warnQuirkOnlyGuesses(st.quirkOnlyGuesses, st.posesFromGuess);
    return st;
}

// pose_graph_builder.h:241-291: one camera + one view per listed image, K = [f 0 w/2; 0 f h/2; 0 0 1], metadata
// "name" / "extension", one pose-graph vertex per image
void PoseGraphBuilder::initializeReconstruction(const size_t& kImageNumber_,
                                                const std::vector<std::tuple<std::string, double, double, double>>& imageData_,
                                                Reconstruction& reconstruction_, PoseGraph& poseGraph_) {
    for (size_t imageIdx = 0; imageIdx < kImageNumber_ && imageIdx < imageData_.size(); ++imageIdx) {
        if (!reconstruction_.addCamera(imageIdx)) continue;
        if (!reconstruction_.addView(imageIdx, imageIdx)) continue;
        const std::string& kImageName = std::get<0>(imageData_[imageIdx]);
        const double kFocalLength = std::get<1>(imageData_[imageIdx]), kImageWidth = std::get<2>(imageData_[imageIdx]),
                     kImageHeight = std::get<3>(imageData_[imageIdx]);
        ViewMetadata& metadata = reconstruction_.getMutableView(imageIdx).getMutableMetadata();
        metadata["name"] = kImageName.size() > 4 ? kImageName.substr(0, kImageName.size() - 4) : kImageName;
        metadata["extension"] = kImageName.size() >= 3 ? kImageName.substr(kImageName.size() - 3) : std::string();
        PinholeCamera& camera = reconstruction_.getMutableCamera(imageIdx);
        camera.setWidth(kImageWidth);
        camera.setHeight(kImageHeight);
        camera.setIntrinsics(kFocalLength, kFocalLength, kImageWidth / 2.0, kImageHeight / 2.0);
        poseGraph_.addVertex(imageIdx);
    }
}

PoseGraphBuilder::FeatureRunStatistics PoseGraphBuilder::processFeatures(const std::vector<ViewFeatures>& owned,
                                                                          std::vector<CandidatePair>& cand,
                                                                          PoseGraph& poseGraph_, size_t waveSize,
                                                                          const SimilarityTable* similarityTable,
                                                                          const MatchLookup* cachedMatches) {
    std::vector<ViewFeaturesRef> views(owned.size());
    for (size_t v = 0; v < owned.size(); ++v) {
        if (owned[v].descriptors.size() != owned[v].size() * (size_t)PGI_DESC_DIM) throw PgiError("processFeatures: descriptors must be n x 128");
        views[v].keypoints = owned[v].keypoints.data();
        views[v].descriptors = owned[v].descriptors.data();
        views[v].n = owned[v].size();
        views[v].focalLength = owned[v].focalLength; views[v].width = owned[v].width; views[v].height = owned[v].height;
    }
    return processFeatures(views, cand, poseGraph_, waveSize, similarityTable, cachedMatches);
}

PoseGraphBuilder::FeatureRunStatistics PoseGraphBuilder::processFeatures(const std::vector<ViewFeaturesRef>& views,
                                                                          std::vector<CandidatePair>& cand,
                                                                          PoseGraph& poseGraph_, size_t waveSize,
                                                                          const SimilarityTable* similarityTable,
                                                                          const MatchLookup* cachedMatches) {
    FeatureRunStatistics st;
    const size_t quirkOnlyAtStart = statistics.getCount("[Pose estimation] Quirk-only guesses");
    pgi_ctx* ctx = engine->get();
    // Rows built from descriptor matches arrive in ascending SNN-ratio order (match_select_kernel sorts by (ratio, row), as
    // feature_utils.h:184-186 does): PROGRESSIVE sampling (pgi_params.sampler = 1) draws the early hypotheses from the best-
    // ranked rows.  Measured on ratio-sorted rows where the iteration cap binds (bench.py, round 6): inlier ratio 0.3 AUC@5
    // 0.93 -> 0.98 at 10 % more edges/s; at the reference's 0.4 px threshold 10 % fewer hypotheses.  On rows without an order
    // (tracklet quick matches, guided matches) it neither helps nor harms (tests/test_gpu_parity.py).  On for feature-level runs
    // unless switched off (setProgressiveSampling); restored when the run ends, however it ends.
    struct SamplerGuard {
        pgi_ctx* ctx;
        uint32_t was;
        bool armed;
        ~SamplerGuard() {
            if (!armed) return;
            pgi_params p;
            if (pgi_get_params(ctx, &p) == PGI_SUCCESS) { p.sampler = was; (void)pgi_set_params(ctx, &p); }
        }
    } samplerGuard{ctx, 0u, false};
    if (progressiveSampling) {
        pgi_params p;
        Engine::check(pgi_get_params(ctx, &p));
        if (p.sampler != 1u) {
            samplerGuard.was = p.sampler;
            samplerGuard.armed = true;
            p.sampler = 1u;
            Engine::check(pgi_set_params(ctx, &p));
        }
    }
    // features resident in HBM for the whole run: keypoints, row-major descriptors (guided matching) and the
    // transposed copy + norms (brute-force matching)
    const size_t V = views.size();
    // (one device block for all views: six allocations per view cost more than the uploads -- a quarter of a 40-view run)
    std::vector<pgi_desc_view> descView(V);
    std::vector<pgi_keypoint_view> kpView(V);
    std::vector<pgi_feature_view> featView(V);
    struct ViewLayout { size_t xy, desc, dt, norm, rm, f16; };
    std::vector<ViewLayout> lay(V);
    size_t arenaBytes = 0;
    auto take = [&](size_t bytes) { const size_t o = arenaBytes; arenaBytes += (bytes + 255) & ~(size_t)255; return o; };
    // the raw part -- what the caller hands over: keypoints and descriptors of every view -- comes first and is contiguous,
    // so that a run of views travels as ONE copy; the arrays derived on the device follow
    for (size_t v = 0; v < V; ++v) {
        const uint32_t n = (uint32_t)views[v].size();
        if (n && (!views[v].keypoints || !views[v].descriptors)) throw PgiError("processFeatures: null feature arrays");
        lay[v].xy = take((size_t)n * 8);
        lay[v].desc = take((size_t)n * PGI_DESC_DIM * 4);
    }
    const size_t rawBytes = arenaBytes;
    for (size_t v = 0; v < V; ++v) {
        const uint32_t n_pad = pgi_desc_padded((uint32_t)views[v].size());
        lay[v].dt = take((size_t)n_pad * PGI_DESC_DIM * 4);
        lay[v].norm = take((size_t)n_pad * 4);
        lay[v].rm = take((size_t)n_pad * PGI_DESC_DIM * 4);
        lay[v].f16 = take((size_t)n_pad * PGI_DESC_DIM * 2);
    }
    const std::chrono::steady_clock::time_point uploadStart = std::chrono::steady_clock::now();
    if (!staging) staging = Staging::shared(engineDevice(engine->get()));
    HIP_OK(hipSetDevice(engineDevice(engine->get())));
    // the process-wide arena and wave pool if nobody else is using them (Staging::featBusy), private memory otherwise; the
    // device is idle on them before the loan ends (every exit path, also an exception's)
    struct Loan {
        std::unique_lock<std::mutex> lock;
        pgi_ctx* ctx;
        ~Loan() { if (lock.owns_lock()) (void)pgi_synchronize(ctx); }
    } loan{std::unique_lock<std::mutex>(staging->featBusy, std::try_to_lock), ctx};
    std::unique_ptr<DevBuf> ownArena;
    if (!loan.lock.owns_lock()) ownArena.reset(new DevBuf(std::max<size_t>(arenaBytes, 256)));
    char* const ab = loan.lock.owns_lock() ? staging->lendArena(std::max<size_t>(arenaBytes, 256)) : ownArena->as<char>();
    if (std::getenv("PGI_PIPELINE_TIMING"))
        std::fprintf(stderr, "[processFeatures] arena of %.2f GB allocated in %.3f s\n", arenaBytes / 1e9,
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - uploadStart).count());
    for (size_t v = 0; v < V; ++v) {
        const uint32_t n = (uint32_t)views[v].size(), n_pad = pgi_desc_padded(n);
        const double f = views[v].focalLength, cx = views[v].width / 2.0, cy = views[v].height / 2.0;
        float* xy = reinterpret_cast<float*>(ab + lay[v].xy);
        float* desc = reinterpret_cast<float*>(ab + lay[v].desc);
        float* dt = reinterpret_cast<float*>(ab + lay[v].dt);
        float* norm = reinterpret_cast<float*>(ab + lay[v].norm);
        float* rm = reinterpret_cast<float*>(ab + lay[v].rm);
        uint16_t* f16 = reinterpret_cast<uint16_t*>(ab + lay[v].f16);
        descView[v] = pgi_desc_view{dt, norm, n, n_pad, rm, f16};
        kpView[v] = pgi_keypoint_view{xy, n, 0, f, f, cx, cy};
        featView[v] = pgi_feature_view{xy, desc, n, 0, f, f, cx, cy, views[v].width, views[v].height};
    }
    {   // Upload (round 4).  The caller's arrays are pageable (cv::Mat / std::vector / numpy), and hipMemcpy's pageable path is
        // erratic: the same 1.39 GB took 0.043 s on one box and 0.32-0.36 s on another.  The features now go
        // through the builder's own page-locked ring: the host team copies a run of views (~64 MB) into a ring buffer, the
        // buffer travels as one asynchronous copy on the copy stream, the per-view preparation kernels of the run wait for
        // it on the engine's stream -- while the team already fills the next buffer.  A view whose arrays are page-locked
        // already (hipHostMalloc / hipHostRegister / pgi_host_register) is copied from where it lies.
        std::lock_guard<std::mutex> stagingBusy(staging->busy);
        staging->init(kCoreNumber ? kCoreNumber : 1);
        auto pageLocked = [](const void* p) {
            hipPointerAttribute_t at{};
            if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
            return at.type == hipMemoryTypeHost;
        };
        const size_t runBytes = (size_t)64 << 20;
        size_t largestRun = 0;
        struct Run { size_t v0, v1; };
        std::vector<Run> runs;
        for (size_t v = 0; v < V;) {
            size_t v1 = v;
            const size_t begin = lay[v].xy;
            auto endOf = [&](size_t w) { return w + 1 < V ? lay[w + 1].xy : rawBytes; };
            while (v1 < V && (v1 == v || endOf(v1) - begin <= runBytes)) ++v1;
            runs.push_back(Run{v, v1});
            largestRun = std::max(largestRun, endOf(v1 - 1) - begin);
            v = v1;
        }
        staging->reserve(staging->active, 0, largestRun, 0);
        hipStream_t copy = staging->copy;
        // (where the host's time goes: printed with PGI_PIPELINE_TIMING, and whenever the stage is far slower than PCIe allows)
        auto nowS = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double tWait = 0, tStage = 0, tPrep = 0;
        for (size_t r = 0; r < runs.size(); ++r) {
            const Run& run = runs[r];
            const int slot = (int)(r % Staging::kRing);
            double tm = nowS();
            if (r >= (size_t)Staging::kRing) HIP_OK(hipEventSynchronize(staging->up[slot]));
            tWait += nowS() - tm;
            tm = nowS();
            char* const hb = (char*)staging->ring[slot];
            const size_t begin = lay[run.v0].xy, end = run.v1 < V ? lay[run.v1].xy : rawBytes;
            // pieces of at most 1 MB, so that the team shares a run evenly whatever the views' sizes
            struct Piece { const char* src; size_t dstOff, bytes; };
            std::vector<Piece> pieces;
            bool direct = true;
            for (size_t v = run.v0; v < run.v1; ++v) direct = direct && (!views[v].size() || (pageLocked(views[v].keypoints) && pageLocked(views[v].descriptors)));
            for (size_t v = run.v0; v < run.v1 && !direct; ++v) {
                const size_t n = views[v].size();
                const char* src[2] = {(const char*)views[v].keypoints, (const char*)views[v].descriptors};
                const size_t bytes[2] = {n * 8, n * (size_t)PGI_DESC_DIM * 4}, dst[2] = {lay[v].xy - begin, lay[v].desc - begin};
                for (int a = 0; a < 2; ++a)
                    for (size_t o = 0; o < bytes[a]; o += (size_t)1 << 20)
                        pieces.push_back(Piece{src[a] + o, dst[a] + o, std::min<size_t>((size_t)1 << 20, bytes[a] - o)});
            }
            if (direct) {
                for (size_t v = run.v0; v < run.v1; ++v) {
                    const size_t n = views[v].size();
                    if (!n) continue;
                    HIP_OK(hipMemcpyAsync(ab + lay[v].xy, views[v].keypoints, n * 8, hipMemcpyHostToDevice, copy));
                    HIP_OK(hipMemcpyAsync(ab + lay[v].desc, views[v].descriptors, n * (size_t)PGI_DESC_DIM * 4, hipMemcpyHostToDevice, copy));
                }
            } else {
                staging->pool->run(pieces.size(), [&](size_t i) { memcpy(hb + pieces[i].dstOff, pieces[i].src, pieces[i].bytes); });
                if (end > begin) HIP_OK(hipMemcpyAsync(ab + begin, hb, end - begin, hipMemcpyHostToDevice, copy));
            }
            HIP_OK(hipEventRecord(staging->up[slot], copy));
            HIP_OK(hipStreamWaitEvent(engineStream(engine->get()), staging->up[slot], 0));  // the stream the engine enqueues on
            tStage += nowS() - tm;
            tm = nowS();
            for (size_t v = run.v0; v < run.v1; ++v) {
                const uint32_t n = (uint32_t)views[v].size();
                Engine::check(pgi_desc_prepare(ctx, featView[v].d_desc, n, const_cast<float*>(descView[v].d_desc_t), const_cast<float*>(descView[v].d_norm)));
                Engine::check(pgi_desc_prepare_screen(ctx, featView[v].d_desc, n, const_cast<float*>(descView[v].d_desc_rm),
                                                      const_cast<uint16_t*>(descView[v].d_desc_f16)));
            }
            tPrep += nowS() - tm;
        }
        double tm = nowS();
        Engine::check(pgi_synchronize(ctx));
        HIP_OK(hipStreamSynchronize(copy));
        const double tDrain = nowS() - tm, all = tWait + tStage + tPrep + tDrain;
        if (std::getenv("PGI_PIPELINE_TIMING") || all > 0.1 + rawBytes / 10.0e9)
            std::fprintf(stderr, "[processFeatures] upload of %.2f GB in %zu runs: waiting for a ring buffer %.3f s, host copies into the ring + copy "
                                 "calls %.3f s, preparation launches %.3f s, waiting for the device at the end %.3f s\n",
                         rawBytes / 1e9, runs.size(), tWait, tStage, tPrep, tDrain);
    }
    Engine::check(pgi_synchronize(ctx));
    st.secUpload = std::chrono::duration<double>(std::chrono::steady_clock::now() - uploadStart).count();
    std::stable_sort(cand.begin(), cand.end(), [](const CandidatePair& a, const CandidatePair& b) {
        if (a.similarity != b.similarity) return a.similarity > b.similarity;
        return std::make_pair(a.src, a.dst) < std::make_pair(b.src, b.dst);
    });
    VisibilityTable visibilityTable(V);  // :366-367
    for (const EdgeId& id : poseGraph_.getEdgeIds()) visibilityTable.addLink(id.first, id.second);
    Tracklets tracks(V);                 // :388; host store, used only when the device store is switched off
    struct DeviceStore {                 // pgi_tracklets_* (csrc/pgi_tracklets.hip): the same bookkeeping in HBM
        pgi_tracklets* t = nullptr;
        ~DeviceStore() { pgi_tracklets_destroy(t); }
    } store;
    const bool deviceTracks = deviceTracklets && kUseEpipolarHashing;
    if (deviceTracks) {
        store.t = pgi_tracklets_create(ctx, (uint32_t)std::max<size_t>(V, 1));
        if (!store.t) throw PgiError(pgi_last_error());
    }
    const bool pathFinding = kUsePathFinding && similarityTable != nullptr;
    uint64_t seed = 0;
    typedef std::vector<Tracklets::Match> Matches;

    typedef std::chrono::steady_clock Clock;
    auto since = [](Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); };
    DevPool ownPool;
    DevPool& pool = loan.lock.owns_lock() ? staging->featPool : ownPool;
    // The path searches of the NEXT wave run on the host while this wave's guided matching occupies the GPU: they read the
    // pose graph and the visibility as this wave's commit leaves them -- exactly what they would see at the start of the
    // next wave -- so the guesses, the counters and the graph are the same as without the overlap.
    struct SearchedWave {
        std::vector<CandidatePair> wave;
        bool searched = false;            // the vectors below are filled (path finding on)
        std::vector<uint8_t> found_pose;  // per pair of `wave`
        std::vector<double> pose;         // 12 per pair
        std::vector<size_t> touched, found;
        double seconds = 0;
    };
    size_t cursor = 0;  // next candidate
    auto formWave = [&](std::vector<CandidatePair>& w) {
        w.clear();
        while (cursor < cand.size() && w.size() < waveSize) {
            const CandidatePair& cp = cand[cursor];
            if (cp.similarity < kSimilarityThreshold) { cursor = cand.size(); break; }
            if (cp.src >= V || cp.dst >= V) throw PgiError("processFeatures: view index out of range");
            ++cursor;
            if (poseGraph_.hasEdge(cp.src, cp.dst) || poseGraph_.hasEdge(cp.dst, cp.src)) continue;  // :426-431
            w.push_back(cp);
        }
    };
    auto searchWave = [&](SearchedWave& sw) {  // findPath (:785-862) for every pair of the wave the graph already connects
        sw.searched = false;
        if (!pathFinding || sw.wave.empty()) return;
        const Clock::time_point t0 = Clock::now();
        const size_t n = sw.wave.size();
        sw.found_pose.assign(n, 0);
        sw.pose.assign(12 * n, 0.0);
        sw.touched.assign(n, 0);
        sw.found.assign(n, 0);
        ImageSimilarityHeuristics heuristics(*similarityTable);
        AStarTraversal<ImageSimilarityHeuristics> traversal(&poseGraph_, heuristics, kTraversalHeuristicsWeight, 0.0, kMaximumSearchDepth);
        traversal.setGraphFrozen(true);  // (commits happen between searchWave calls, never during one)
        // small waves stay on the calling thread: starting threads costs more than a few hundred short searches
        parallelFor(n, n >= 2048 ? (kCoreNumber ? kCoreNumber : 1) : 1, [&](size_t i) {
            if (!visibilityTable.hasLink(sw.wave[i].src, sw.wave[i].dst)) return;
            std::vector<ViewId> path;
            std::vector<SE3d> poses;
            bool exists = false;
            traversal.getPath(sw.wave[i].src, sw.wave[i].dst, path, poses, sw.touched[i], sw.found[i], exists);
            if (poses.empty()) return;
            for (int c = 0; c < 9; ++c) sw.pose[12 * i + c] = poses.back().R[c];
            for (int c = 0; c < 3; ++c) sw.pose[12 * i + 9 + c] = poses.back().t[c];
            sw.found_pose[i] = 1;
        });
        sw.searched = true;
        sw.seconds = since(t0);
    };
    auto processWave = [&](const SearchedWave& current, SearchedWave& next) {
        const std::vector<CandidatePair>& wave = current.wave;
        const size_t P = wave.size();
        if (!P) return;
        Engine::check(pgi_synchronize(ctx));  // the previous wave's kernels are done with the pool
        pool.reset();
        Clock::time_point tick = Clock::now();
        // (1) quick matching from tracklets for pairs the graph already connects (:493-518)
        std::vector<Matches> matches(P);
        std::vector<size_t> matchCount(P, 0);  // matches[i].size() where the list itself never reaches the host
        std::vector<char> quick(P, 0), visible(P, 0), fromHost(P, 0);
        for (size_t i = 0; i < P; ++i) visible[i] = visibilityTable.hasLink(wave[i].src, wave[i].dst);
        // device store: one batched query for every visible pair; the rows stay in HBM
        const uint32_t qstride = (uint32_t)kMaximumTrackletNumber + 1;
        std::vector<size_t> queryOf(P, (size_t)-1);
        std::vector<uint32_t> qcnt;
        std::unique_ptr<WaveBuf> qsrc, qdst, qcntDev;
        if (deviceTracks) {
            std::vector<uint32_t> qs, qd;
            for (size_t i = 0; i < P; ++i)
                if (visible[i]) { queryOf[i] = qs.size(); qs.push_back((uint32_t)wave[i].src); qd.push_back((uint32_t)wave[i].dst); }
            if (!qs.empty()) {
                const size_t Q = qs.size();
                qsrc.reset(new WaveBuf(pool, Q * (size_t)qstride * 4));
                qdst.reset(new WaveBuf(pool, Q * (size_t)qstride * 4));
                qcntDev.reset(new WaveBuf(pool, Q * 4));
                Engine::check(pgi_tracklets_get_batch(store.t, qs.data(), qd.data(), (uint32_t)Q, (uint32_t)kMaximumTrackletNumber, qstride,
                                                      qsrc->as<uint32_t>(), qdst->as<uint32_t>(), qcntDev->as<uint32_t>()));
                Engine::check(pgi_synchronize(ctx));
                qcnt.resize(Q);
                d2h(qcnt.data(), qcntDev->p, Q * 4);
            }
        }
        for (size_t i = 0; i < P; ++i) {
            if (kUseEpipolarHashing && visible[i]) {
                if (deviceTracks) {
                    matchCount[i] = qcnt[queryOf[i]];
                } else {
                    tracks.getCorrespondences(matches[i], wave[i].src, wave[i].dst, kMaximumTrackletNumber);
                    matchCount[i] = matches[i].size();
                }
                if (matchCount[i] < kMinimumInlierNumber) { matches[i].clear(); matchCount[i] = 0; }
                else { quick[i] = fromHost[i] = 1; ++st.quickMatchingRuns; }
            }
            if (!quick[i] && cachedMatches && (*cachedMatches)(wave[i].src, wave[i].dst, matches[i])) {  // feature_utils.h:115-133
                fromHost[i] = 1;
                matchCount[i] = matches[i].size();
                ++st.cachedMatchLoads;
            }
        }
        {   // :505-518
            const double dt = since(tick);
            size_t tried = 0, ok = 0;
            for (size_t i = 0; i < P; ++i) { tried += (kUseEpipolarHashing && visible[i]) ? 1 : 0; ok += quick[i] ? 1 : 0; }
            st.secQuickMatching += dt;
            if (tried) statistics.addTime("[Quick matching]", dt, tried);
            if (ok) statistics.addCount("[Quick matching] Runs", ok, ok);
        }
        tick = Clock::now();
        // batch order: descriptor-matched pairs first (the matcher writes rows 0..Pn-1), host-provided matches after them
        std::vector<size_t> order;
        for (size_t i = 0; i < P; ++i) if (!fromHost[i]) order.push_back(i);
        const size_t Pn = order.size();
        for (size_t i = 0; i < P; ++i) if (fromHost[i]) order.push_back(i);
        uint32_t mm = 1;
        for (size_t k = 0; k < P; ++k) {
            const size_t i = order[k];
            mm = std::max(mm, fromHost[i] ? (uint32_t)matchCount[i] : (uint32_t)views[wave[i].src].size());
        }
        WaveBuf dsrc(pool, P * (size_t)mm * 4), ddst(pool, P * (size_t)mm * 4), dratio(pool, P * (size_t)mm * 8), dcnt(pool, P * 4);
        std::vector<uint32_t> hsrc(P * (size_t)mm), hdst(P * (size_t)mm), hcnt(P, 0);
        // (2) descriptor matching for the others, one launch sequence (:521-546)
        if (Pn) {
            std::vector<pgi_desc_view> a(Pn), b(Pn);
            for (size_t k = 0; k < Pn; ++k) { a[k] = descView[wave[order[k]].src]; b[k] = descView[wave[order[k]].dst]; }
            Engine::check(pgi_match_descriptors_batch(ctx, a.data(), b.data(), (uint32_t)Pn, mm, dsrc.as<uint32_t>(), ddst.as<uint32_t>(),
                                                      dratio.as<double>(), dcnt.as<uint32_t>()));
            Engine::check(pgi_synchronize(ctx));
            d2h(hcnt.data(), dcnt.p, Pn * 4);
            st.matchingRuns += Pn;
            for (size_t k = 0; k < Pn; ++k) matchCount[order[k]] = hcnt[k];
            if (kUseEpipolarHashing && !deviceTracks) {  // the host store consumes the match lists themselves
                d2h(hsrc.data(), dsrc.p, Pn * (size_t)mm * 4);
                d2h(hdst.data(), ddst.p, Pn * (size_t)mm * 4);
                for (size_t k = 0; k < Pn; ++k) {
                    Matches& m = matches[order[k]];
                    m.resize(hcnt[k]);
                    for (uint32_t q = 0; q < hcnt[k]; ++q) m[q] = Tracklets::Match(hsrc[k * (size_t)mm + q], hdst[k * (size_t)mm + q], 0.0);
                }
            }
        }
        {   // :545-546
            const double dt = since(tick);
            st.secMatching += dt;
            if (Pn) { statistics.addTime("[Matching]", dt, Pn); statistics.addCount("[Matching] Runs", Pn, Pn); }
        }
        tick = Clock::now();
        bool hostRows = false;
        for (size_t k = Pn; k < P; ++k) {  // tracklet / cached matches join the same device layout
            const size_t i = order[k];
            hcnt[k] = (uint32_t)matchCount[i];
            if (deviceTracks && quick[i]) {  // rows of the batched query: device to device
                // consecutive quick pairs sit in consecutive rows of both tables: one strided copy per run instead of two
                // copies per pair (config 3 at full size issued 12 000 of them per run: 29 ms of device time, more on the host)
                const size_t q0 = queryOf[i];
                size_t run = 1;
                uint32_t widest = hcnt[k];
                while (k + run < P && quick[order[k + run]] && queryOf[order[k + run]] == q0 + run) {
                    hcnt[k + run] = (uint32_t)matchCount[order[k + run]];
                    widest = std::max(widest, hcnt[k + run]);
                    ++run;
                }
                const size_t width = (size_t)std::min<uint32_t>(widest, std::min<uint32_t>(mm, (uint32_t)qstride)) * 4;
                if (width &&
                    (hipMemcpy2DAsync((uint32_t*)dsrc.p + k * (size_t)mm, (size_t)mm * 4, qsrc->as<uint32_t>() + q0 * (size_t)qstride,
                                      (size_t)qstride * 4, width, run, hipMemcpyDeviceToDevice, nullptr) != hipSuccess ||
                     hipMemcpy2DAsync((uint32_t*)ddst.p + k * (size_t)mm, (size_t)mm * 4, qdst->as<uint32_t>() + q0 * (size_t)qstride,
                                      (size_t)qstride * 4, width, run, hipMemcpyDeviceToDevice, nullptr) != hipSuccess))
                    throw PgiError("hipMemcpy2D D2D failed");
                k += run - 1;
                continue;
            }
            const Matches& m = matches[i];
            hostRows = true;
            for (size_t q = 0; q < m.size(); ++q) {
                hsrc[k * (size_t)mm + q] = (uint32_t)std::get<0>(m[q]);
                hdst[k * (size_t)mm + q] = (uint32_t)std::get<1>(m[q]);
            }
        }
        std::vector<char> skipped(P, 0);
        for (size_t k = 0; k < P; ++k)
            if (hcnt[k] < kMinimumPointNumber) { skipped[k] = 1; hcnt[k] = 0; ++st.tooFewMatches; }  // :550-551 `continue`
        if (hostRows) {
            for (size_t k = Pn; k < P; ++k) {
                if (deviceTracks && quick[order[k]]) continue;
                h2d((uint32_t*)dsrc.p + k * (size_t)mm, hsrc.data() + k * (size_t)mm, (size_t)matchCount[order[k]] * 4);
                h2d((uint32_t*)ddst.p + k * (size_t)mm, hdst.data() + k * (size_t)mm, (size_t)matchCount[order[k]] * 4);
            }
        }
        h2d(dcnt.p, hcnt.data(), P * 4);
        // (3) createCorrespondenceMatrix on the device (:553-565)
        std::vector<pgi_keypoint_view> ka(P), kb(P);
        for (size_t k = 0; k < P; ++k) { ka[k] = kpView[wave[order[k]].src]; kb[k] = kpView[wave[order[k]].dst]; }
        const size_t cap = P * (size_t)mm;
        WaveBuf dx1(pool, cap * 4), dy1(pool, cap * 4), dx2(pool, cap * 4), dy2(pool, cap * 4), doff(pool, (P + 1) * 8), dthr(pool, P * 8),
            dguess(pool, P * 96), dhas(pool, P), dedges(pool, P * sizeof(pgi_edge)), dmasks(pool, cap);
        Engine::check(pgi_build_correspondences(ctx, ka.data(), kb.data(), (uint32_t)P, mm, dsrc.as<uint32_t>(), ddst.as<uint32_t>(),
                                                dcnt.as<uint32_t>(), 0, kInlierOutlierThreshold, 0, dx1.as<float>(), dy1.as<float>(),
                                                dx2.as<float>(), dy2.as<float>(), doff.as<uint64_t>(), dthr.as<double>()));
        pgi_batch b{};
        b.d_x1 = dx1.as<float>(); b.d_y1 = dy1.as<float>(); b.d_x2 = dx2.as<float>(); b.d_y2 = dy2.as<float>();
        b.d_offsets = doff.as<uint64_t>(); b.d_thr = dthr.as<double>();
        b.n_pairs = (uint32_t)P; b.max_corr = mm; b.pair_id_base = 0; b.seed = seed++;
        Engine::check(pgi_synchronize(ctx));
        st.secCorrespondences += since(tick); tick = Clock::now();
        // (4) A* pose guesses on the graph committed by earlier waves (:568-599), screened in one launch (:798-811)
        std::vector<double> guess(12 * P, 0.0);
        std::vector<uint8_t> has(P, 0);
        bool anyGuess = false;
        size_t waveSearched = 0, waveTouched = 0, waveFound = 0;
        double searchSeconds = 0;  // (spent during the previous wave's guided matching when the searches ran ahead)
        if (pathFinding) {
            SearchedWave inlineSearch;
            const SearchedWave* sw = &current;
            if (!current.searched) {  // the first wave, or a caller without the overlap
                inlineSearch.wave = wave;
                searchWave(inlineSearch);
                sw = &inlineSearch;
            }
            searchSeconds = sw->seconds;
            for (size_t k = 0; k < P; ++k) {
                const size_t i = order[k];
                if (!visible[i] || skipped[k]) continue;  // (a pair without enough matches never asked for a path, :550-551)
                ++st.pathsSearched;
                st.touchedNodes += sw->touched[i];
                st.pathsFound += sw->found[i];
                ++waveSearched;
                waveTouched += sw->touched[i];
                waveFound += sw->found[i];
                if (!sw->found_pose[i]) continue;
                for (int c = 0; c < 12; ++c) guess[12 * k + c] = sw->pose[12 * i + c];
                has[k] = 1;
                anyGuess = true;
            }
        }
        {
            const double dt = (current.searched ? 0.0 : since(tick)) + (current.searched ? searchSeconds : 0.0);
            st.secAStar += dt;
            if (waveSearched) {  // :599-602
                statistics.addTime("[A*]", dt, waveSearched);
                statistics.addCount("[A*] Runs", waveSearched, waveSearched);
                statistics.addCount("[A*] Touched nodes", waveTouched, waveSearched);
                statistics.addCount("[A*] Paths tested", waveFound, waveSearched);
            }
        }
        tick = Clock::now();
        std::vector<uint8_t> screened;      // pairs that carried a chained pose into the screening launch
        std::vector<uint32_t> guessInliers; // their inlier counts under the squared bound (1.5 thr)^2
        if (anyGuess && rotationGuidedGuesses) {  // guess_mode 1 scores its own two-point hypotheses: no 5-inlier screening
            h2d(dguess.p, guess.data(), P * 96);
            h2d(dhas.p, has.data(), P);
            b.d_guess_Rt = dguess.as<double>();
            b.d_has_guess = dhas.as<uint8_t>();
        } else if (anyGuess) {
            screened = has;
            std::vector<double> thr(P), Eg(9 * P, 0.0), tau2(P, 0.0);
            Engine::check(pgi_synchronize(ctx));
            d2h(thr.data(), dthr.p, P * 8);
            for (size_t k = 0; k < P; ++k) {
                if (!has[k]) { Eg[9 * k] = 1.0; continue; }
                SE3d g;
                for (int c = 0; c < 9; ++c) g.R[c] = guess[12 * k + c];
                for (int c = 0; c < 3; ++c) g.t[c] = guess[12 * k + 9 + c];
                const Matrix3d E = pose::getEssentialMatrixFromRelativePose(g);
                for (int c = 0; c < 9; ++c) Eg[9 * k + c] = E[c];
                tau2[k] = (1.5 * thr[k]) * (1.5 * thr[k]);
            }
            WaveBuf dE(pool, P * 72), dtau(pool, P * 8), dscore(pool, P * 4);
            h2d(dE.p, Eg.data(), P * 72);
            h2d(dtau.p, tau2.data(), P * 8);
            Engine::check(pgi_score_pose_batch(ctx, &b, dE.as<double>(), dtau.as<double>(), dscore.as<uint32_t>(), nullptr));
            Engine::check(pgi_synchronize(ctx));
            guessInliers.resize(P);
            d2h(guessInliers.data(), dscore.p, P * 4);
            for (size_t k = 0; k < P; ++k)
                if (has[k] && guessInliers[k] < 5) has[k] = 0;
            h2d(dguess.p, guess.data(), P * 96);
            h2d(dhas.p, has.data(), P);
            b.d_guess_Rt = dguess.as<double>();
            b.d_has_guess = dhas.as<uint8_t>();
        }
        // (5) estimatePose for the whole wave (:616-627)
        Engine::check(pgi_estimate_pose_batch(ctx, &b, dedges.as<pgi_edge>(), dmasks.as<uint8_t>()));
        Engine::check(pgi_synchronize(ctx));
        std::vector<pgi_edge> edges(P);
        std::vector<uint64_t> off(P + 1);
        d2h(edges.data(), dedges.p, P * sizeof(pgi_edge));
        d2h(off.data(), doff.p, (P + 1) * 8);
        std::vector<uint8_t> masks((size_t)off[P]);
        d2h(masks.data(), dmasks.p, masks.size());
        {   // :636-638
            const double dt = since(tick);
            st.secPoseEstimation += dt;
            size_t runs = 0, inl = 0;
            for (size_t k = 0; k < P; ++k)
                if (!skipped[k]) { ++runs; inl += edges[k].n_inl; }
            if (runs) {
                statistics.addTime("[Pose estimation]", dt, runs);
                statistics.addCount("[Pose estimation] Runs", runs, runs);
                statistics.addCount("[Pose estimation] Inlier number", inl, runs);
            }
            if (!screened.empty()) {  // accepted guesses that only the un-squared getInliers bound let through (see estimatePoses)
                size_t quirkOnly = 0;
                for (size_t k = 0; k < P; ++k)
                    if (screened[k] && edges[k].used_guess && edges[k].status == PGI_EDGE_OK && guessInliers[k] < kMinimumInlierNumber)
                        ++quirkOnly;
                statistics.addCount("[Pose estimation] Quirk-only guesses", quirkOnly, runs);
            }
        }
        tick = Clock::now();
        tick = Clock::now();
        // (6b) guided matching for the successful tracklet pairs, one launch sequence (:657-686)
        std::vector<size_t> guidedOf;
        std::vector<pgi_feature_view> ga, gb;
        std::vector<double> gpose;
        for (size_t k = Pn; k < P; ++k) {
            if (skipped[k] || edges[k].status != PGI_EDGE_OK || !quick[order[k]]) continue;
            guidedOf.push_back(k);
            ga.push_back(featView[wave[order[k]].src]);
            gb.push_back(featView[wave[order[k]].dst]);
            for (int c = 0; c < 9; ++c) gpose.push_back(edges[k].R[c]);
            for (int c = 0; c < 3; ++c) gpose.push_back(edges[k].t[c]);
        }
        const uint32_t gstride = (uint32_t)std::max<size_t>(1, kMaximumPointNumberForEpipolarHashing);
        constexpr uint32_t kEpipolarBins = 45;  // HashingBasedMatcherWithPose<false, 45> (pose_graph_builder.h:738)
        std::vector<uint32_t> gsrc, gdst, gcnt;
        std::unique_ptr<WaveBuf> ds, dd, dr, dc;
        if (!guidedOf.empty()) {
            const size_t G = guidedOf.size();
            ds.reset(new WaveBuf(pool, G * (size_t)gstride * 4)); dd.reset(new WaveBuf(pool, G * (size_t)gstride * 4));
            dr.reset(new WaveBuf(pool, G * (size_t)gstride * 8)); dc.reset(new WaveBuf(pool, G * 4));
            Engine::check(pgi_guided_match_batch(ctx, ga.data(), gb.data(), (uint32_t)G, gpose.data(), kEpipolarBins, gstride, gstride, ds->as<uint32_t>(),
                                                 dd->as<uint32_t>(), dr->as<double>(), dc->as<uint32_t>()));
        }
        // while the GPU works on that: (6a) the edges in wave order (:645-654) and the visibility (:692) -- nothing of it depends
        // on the guided matches, and the next wave's path searches need it --
        std::vector<size_t> slotOf(P);
        for (size_t k = 0; k < P; ++k) slotOf[order[k]] = k;
        double commitSeconds = 0;
        {
            const Clock::time_point t0 = Clock::now();
            for (size_t i = 0; i < P; ++i) {
                const size_t k = slotOf[i];
                ++st.pairsProcessed;
                if (skipped[k]) continue;
                st.hypotheses += edges[k].iters;
                st.posesFromGuess += edges[k].used_guess;
                if (edges[k].status != PGI_EDGE_OK) continue;  // :641-642
                SE3d T;
                for (int c = 0; c < 9; ++c) T.R[c] = edges[k].R[c];
                for (int c = 0; c < 3; ++c) T.t[c] = edges[k].t[c];
                poseGraph_.addVertex(wave[i].src);
                poseGraph_.addVertex(wave[i].dst);
                poseGraph_.addEdge(wave[i].src, wave[i].dst, Pose(T), (double)edges[k].n_inl / (double)matchCount[i]);
                ++st.edgesAdded;
                visibilityTable.addLink(wave[i].src, wave[i].dst);
            }
            commitSeconds = since(t0);
        }
        // ... then the next wave and its path searches
        formWave(next.wave);
        searchWave(next);
        if (!guidedOf.empty()) {
            const size_t G = guidedOf.size();
            Engine::check(pgi_synchronize(ctx));
            gcnt.resize(G);
            d2h(gcnt.data(), dc->p, G * 4);
            if (!deviceTracks) {
                gsrc.resize(G * (size_t)gstride); gdst.resize(G * (size_t)gstride);
                d2h(gsrc.data(), ds->p, gsrc.size() * 4);
                d2h(gdst.data(), dd->p, gdst.size() * 4);
            }
            st.guidedMatchingRuns += G;
        }
        {   // :684-686 (the commit and the searches that ran meanwhile are booked under their own stages, not here)
            const double dt = std::max(0.0, since(tick) - next.seconds - commitSeconds);
            st.secGuidedMatching += dt;
            if (!guidedOf.empty()) {
                size_t extra = 0;
                for (uint32_t c : gcnt) extra += c;
                statistics.addTime("[Epipolar Hashing]", dt, guidedOf.size());
                statistics.addCount("[Epipolar Hashing] Runs", guidedOf.size(), guidedOf.size());
                statistics.addCount("[Epipolar Hashing] Correspondences added", extra, guidedOf.size());
            }
        }
        tick = Clock::now();
        // (7) the tracklets of the committed edges, in wave order (:677-681, :702-709)
        std::vector<size_t> guidedSlot(P, (size_t)-1);
        for (size_t q = 0; q < guidedOf.size(); ++q) guidedSlot[guidedOf[q]] = q;
        std::vector<pgi_tracklet_pair> adds;
        for (size_t i = 0; i < P && kUseEpipolarHashing; ++i) {
            const size_t k = slotOf[i];
            if (skipped[k] || edges[k].status != PGI_EDGE_OK) continue;
            if (deviceTracks) {  // the add() calls of the wave, in commit order, straight from the device rows
                pgi_tracklet_pair a{};
                a.view_src = (uint32_t)wave[i].src;
                a.view_dst = (uint32_t)wave[i].dst;
                if (quick[i]) {
                    const size_t q = guidedSlot[k];
                    a.n_max = gcnt[q];
                    a.d_src = ds->as<uint32_t>() + q * (size_t)gstride;
                    a.d_dst = dd->as<uint32_t>() + q * (size_t)gstride;
                    st.guidedMatchesAdded += gcnt[q];
                } else {
                    a.n_max = hcnt[k];
                    a.d_src = dsrc.as<uint32_t>() + k * (size_t)mm;
                    a.d_dst = ddst.as<uint32_t>() + k * (size_t)mm;
                    a.d_mask = dmasks.as<uint8_t>() + off[k];
                }
                adds.push_back(a);
            } else if (quick[i]) {
                const size_t q = guidedSlot[k];
                Matches extra(gcnt[q]);
                for (uint32_t r = 0; r < gcnt[q]; ++r) extra[r] = Tracklets::Match(gsrc[q * (size_t)gstride + r], gdst[q * (size_t)gstride + r], 0.0);
                tracks.add(wave[i].src, wave[i].dst, extra, std::vector<uchar>(extra.size(), 1));
                st.guidedMatchesAdded += extra.size();
            } else {
                const std::vector<uchar> mask(masks.begin() + (size_t)off[k], masks.begin() + (size_t)off[k + 1]);
                tracks.add(wave[i].src, wave[i].dst, matches[i], mask);
            }
        }
        if (!adds.empty()) Engine::check(pgi_tracklets_add_batch(store.t, adds.data(), (uint32_t)adds.size()));
        {   // :698-699 (the edge / visibility commit above is part of it)
            const double dt = since(tick) + commitSeconds;
            st.secTrackUpdate += dt;
            statistics.addTime("[Visibility update]", dt, P);
            statistics.addCount("[Visibility update] Runs", P, P);
        }
        ++st.waves;
    };

    SearchedWave current, next;
    formWave(current.wave);  // (the first wave's searches, if any, run inside processWave)
    while (!current.wave.empty()) {
        next = SearchedWave();
        processWave(current, next);
        current = std::move(next);
    }
    warnQuirkOnlyGuesses(statistics.getCount("[Pose estimation] Quirk-only guesses") - quirkOnlyAtStart, st.posesFromGuess);
    st.trackNumber = tracks.trackNumber();
    if (deviceTracks) {
        uint64_t n = 0;
        Engine::check(pgi_tracklets_info(store.t, &n, nullptr, nullptr));
        st.trackNumber = (size_t)n;
    }
    return st;
}

namespace pose {
// pose_utils.h:172-252 -- one re-entrant call on host pointers (pgi_pose_from_essential_host: private slot of the context,
// no allocation once warm); every row votes, as in the reference (:203)
int getPoseFromEssentialMatrix(Engine& eng, const Matrix3d& E, const CorrespondenceMatrix& c, Matrix3d& rotation_,
                               Vector3d& translation_) {
    uint32_t votes = 0;
    Engine::check(pgi_pose_from_essential_host(eng.get(), E.data(), c.rows ? c.ptr() : nullptr, (uint32_t)c.rows, nullptr,
                                               rotation_.data(), translation_.data(), &votes, nullptr));
    return (int)votes;
}
}  // namespace pose

}  // namespace reconstruction
