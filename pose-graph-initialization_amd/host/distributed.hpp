// distributed.hpp -- one process per GPU: launch environment, bootstrap and the pair partition.
//
// The reference is a single process whose parallelism is the OpenMP loop of pose_graph_builder.h:391-413; the
// MI355X design shards the image pairs over the GPUs of a node instead and gathers the per-edge records once
// (pgi_allgather_edges, include/pgi.h).  This header is the host-side plumbing for that, in C++ like the rest
// of the host layer (no Python needed):
//   LaunchEnv   RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT as exported by torch.distributed.run
//               (or any launcher that sets the same variables)
//   HostComm    a small TCP star (rank 0 <-> every other rank) used (i) to ship the RCCL unique id, (ii) for the
//               scheduler's host-side bookkeeping exchanges and (iii) as the records' transport when ranks share a
//               device (RCCL refuses duplicate devices: the one-GPU test box) -- never on the production data path
//   attach()    wires a transport into an Engine: RCCL over xGMI when every rank owns its own device, host otherwise
//   shardBounds contiguous blocks of the pair list, balanced by total row count
#pragma once
#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "pose_graph_builder.hpp"

namespace reconstruction {
namespace dist {

struct LaunchEnv {
    uint32_t world = 1, rank = 0, localRank = 0;
    std::string masterAddr = "127.0.0.1";
    int masterPort = 29500;
    static LaunchEnv fromEnvironment();
};

class HostComm {
   public:
    // rank 0 listens on masterPort + portOffset (the launcher's own store owns masterPort itself)
    // timeoutSeconds: rendezvous (connect / accept); ioTimeoutSeconds: any later receive (a wave of config 5 keeps a rank
    // busy for well under a second; an hour only guards against a peer that is gone) -- env PGI_COMM_TIMEOUT overrides it
    explicit HostComm(const LaunchEnv& env, int portOffset = 41, double timeoutSeconds = 120.0);
    ~HostComm();
    HostComm(const HostComm&) = delete;
    HostComm& operator=(const HostComm&) = delete;
    uint32_t world() const { return env.world; }
    uint32_t rank() const { return env.rank; }
    const LaunchEnv& launchEnv() const { return env; }
    void broadcast(void* buf, size_t bytes);  // from rank 0
    // every rank contributes bytes[rank]; recv holds the blocks in rank order
    void allgatherv(const void* send, uint64_t sendBytes, void* recv, const uint64_t* bytes);
    template <class T>
    std::vector<T> allgather(const T& mine) {  // one fixed-size record per rank
        std::vector<T> all(env.world);
        std::vector<uint64_t> b(env.world, sizeof(T));
        allgatherv(&mine, sizeof(T), all.data(), b.data());
        return all;
    }
    void barrier();
    // pgi_allgatherv_fn for pgi_comm_init_host; `user` is the HostComm
    static int transportCallback(void* user, const void* send, uint64_t sendBytes, void* recv, const uint64_t* bytes,
                                 uint32_t world);

   protected:
    LaunchEnv env;
    double ioTimeoutSeconds = 3600.0;
    std::vector<int> peers;  // rank 0: socket of rank r at [r]; other ranks: [0] = socket to rank 0
    int listener = -1;
};

enum class Transport { Auto, Rccl, Host };
// hipSetDevice(localRank % visible devices); returns the device index.  Call before constructing the Engine.
int selectDevice(const LaunchEnv& env);
// Installs the exchange in the engine's context.  Auto: RCCL iff all ranks report distinct devices (PCI bus ids),
// unless the environment says PGI_COMM=host / PGI_COMM=rccl.  Returns the transport in use.
Transport attach(Engine& engine, HostComm& comm, Transport want = Transport::Auto);

// Contiguous blocks [lo, hi) of the pair list with balanced total row counts (same rule as
// pyposegraphbuilder.distributed.shard_bounds): block boundaries at the first pair whose prefix sum reaches r/world.
inline std::vector<std::pair<size_t, size_t>> shardBounds(const std::vector<uint64_t>& rowsPerPair, uint32_t world);

// ---- implementation (header-inline: no HIP, so the host-only tests can use it with plain g++) ---------------------
namespace detail {
inline const char* envOr(const char* name, const char* fallback) {
    const char* v = std::getenv(name);
    return (v && *v) ? v : fallback;
}
inline void sendAll(int fd, const void* buf, size_t n) {
    const char* p = (const char*)buf;
    while (n) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k <= 0) throw PgiError("HostComm: send failed");
        p += k;
        n -= (size_t)k;
    }
}
inline void recvAll(int fd, void* buf, size_t n) {
    char* p = (char*)buf;
    while (n) {
        const ssize_t k = ::recv(fd, p, n, 0);
        if (k <= 0) throw PgiError(k < 0 && (errno == EAGAIN || errno == EWOULDBLOCK) ? "HostComm: timed out waiting for a peer"
                                                                                         : "HostComm: peer closed the connection");
        p += k;
        n -= (size_t)k;
    }
}
inline void noDelay(int fd) {
    int one = 1;
    (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
}
// A peer that died must not block the others for ever: receives (and accept on the listener) give up after `seconds`
// (recv / accept then fail with EAGAIN and the caller throws).
inline void recvTimeout(int fd, double seconds) {
    timeval tv{};
    tv.tv_sec = (long)seconds;
    tv.tv_usec = (long)((seconds - (double)tv.tv_sec) * 1e6);
    (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
}
}  // namespace detail

inline LaunchEnv LaunchEnv::fromEnvironment() {
    LaunchEnv e;
    e.world = (uint32_t)std::atoi(detail::envOr("WORLD_SIZE", "1"));
    e.rank = (uint32_t)std::atoi(detail::envOr("RANK", "0"));
    e.localRank = (uint32_t)std::atoi(detail::envOr("LOCAL_RANK", detail::envOr("RANK", "0")));
    e.masterAddr = detail::envOr("MASTER_ADDR", "127.0.0.1");
    e.masterPort = std::atoi(detail::envOr("MASTER_PORT", "29500"));
    if (e.world == 0 || e.rank >= e.world) throw PgiError("LaunchEnv: RANK / WORLD_SIZE are inconsistent");
    return e;
}

inline HostComm::HostComm(const LaunchEnv& env_, int portOffset, double timeoutSeconds) : env(env_) {
    if (env.world <= 1) return;
    if (const char* t = std::getenv("PGI_COMM_TIMEOUT")) ioTimeoutSeconds = std::max(1.0, std::atof(t));
    const int port = env.masterPort + portOffset;
    if (env.rank == 0) {
        listener = ::socket(AF_INET, SOCK_STREAM, 0);
        if (listener < 0) throw PgiError("HostComm: socket failed");
        int one = 1;
        (void)setsockopt(listener, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in a{};
        a.sin_family = AF_INET;
        a.sin_addr.s_addr = htonl(INADDR_ANY);
        a.sin_port = htons((uint16_t)port);
        if (::bind(listener, (sockaddr*)&a, sizeof a) != 0 || ::listen(listener, (int)env.world) != 0)
            throw PgiError("HostComm: cannot listen on port " + std::to_string(port));
        peers.assign(env.world, -1);
        detail::recvTimeout(listener, timeoutSeconds);  // accept() honours SO_RCVTIMEO
        for (uint32_t k = 1; k < env.world; ++k) {
            const int fd = ::accept(listener, nullptr, nullptr);
            if (fd < 0) throw PgiError("HostComm: rank 0 timed out waiting for " + std::to_string(env.world - k) + " rank(s) to connect");
            detail::noDelay(fd);
            detail::recvTimeout(fd, ioTimeoutSeconds);
            uint32_t r = 0;
            detail::recvAll(fd, &r, 4);
            if (r == 0 || r >= env.world || peers[r] >= 0) throw PgiError("HostComm: unexpected rank in handshake");
            peers[r] = fd;
        }
    } else {
        addrinfo hints{}, *res = nullptr;
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        if (getaddrinfo(env.masterAddr.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res)
            throw PgiError("HostComm: cannot resolve " + env.masterAddr);
        const auto t0 = std::chrono::steady_clock::now();
        int fd = -1;
        for (;;) {  // rank 0 may not be listening yet
            fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
                // A connect() to a local port nobody listens on yet can succeed against ITSELF when the kernel hands out that
                // very port as the source (TCP simultaneous open; the port then is taken and rank 0 cannot listen on it --
                // seen once in the suite with eight ranks retrying): such a socket is dropped and the attempt repeated.
                sockaddr_in me{}, peer{};
                socklen_t lm = sizeof me, lp = sizeof peer;
                const bool self = ::getsockname(fd, (sockaddr*)&me, &lm) == 0 && ::getpeername(fd, (sockaddr*)&peer, &lp) == 0 &&
                                  me.sin_port == peer.sin_port && me.sin_addr.s_addr == peer.sin_addr.s_addr;
                if (!self) break;
            }
            if (fd >= 0) ::close(fd);
            fd = -1;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
        freeaddrinfo(res);
        if (fd < 0) throw PgiError("HostComm: cannot reach rank 0 at " + env.masterAddr + ":" + std::to_string(port));
        detail::noDelay(fd);
        detail::recvTimeout(fd, ioTimeoutSeconds);
        detail::sendAll(fd, &env.rank, 4);
        peers.assign(1, fd);
    }
}

inline HostComm::~HostComm() {
    for (int fd : peers)
        if (fd >= 0) ::close(fd);
    if (listener >= 0) ::close(listener);
}

inline void HostComm::broadcast(void* buf, size_t bytes) {
    if (env.world <= 1 || bytes == 0) return;
    if (env.rank == 0) {
        for (uint32_t r = 1; r < env.world; ++r) detail::sendAll(peers[r], buf, bytes);
    } else {
        detail::recvAll(peers[0], buf, bytes);
    }
}

inline void HostComm::allgatherv(const void* send, uint64_t sendBytes, void* recv, const uint64_t* bytes) {
    std::vector<uint64_t> off((size_t)env.world + 1, 0);
    for (uint32_t r = 0; r < env.world; ++r) off[r + 1] = off[r] + bytes[r];
    if (sendBytes != bytes[env.rank]) throw PgiError("HostComm::allgatherv: block size mismatch");
    char* out = (char*)recv;
    if (sendBytes && (const char*)send != out + off[env.rank]) std::memcpy(out + off[env.rank], send, sendBytes);
    if (env.world <= 1) return;
    if (env.rank == 0) {
        for (uint32_t r = 1; r < env.world; ++r)
            if (bytes[r]) detail::recvAll(peers[r], out + off[r], bytes[r]);
        for (uint32_t r = 1; r < env.world; ++r)
            if (off[env.world]) detail::sendAll(peers[r], out, off[env.world]);
    } else {
        if (sendBytes) detail::sendAll(peers[0], send, sendBytes);
        if (off[env.world]) detail::recvAll(peers[0], out, off[env.world]);
    }
}

inline void HostComm::barrier() {
    const uint8_t token = 1;
    (void)allgather(token);
}

inline int HostComm::transportCallback(void* user, const void* send, uint64_t sendBytes, void* recv, const uint64_t* bytes,
                                uint32_t world) {
    HostComm* c = static_cast<HostComm*>(user);
    if (!c || world != c->world()) return 1;
    try {
        c->allgatherv(send, sendBytes, recv, bytes);
    } catch (const std::exception&) {
        return 2;
    }
    return 0;
}

inline std::vector<std::pair<size_t, size_t>> shardBounds(const std::vector<uint64_t>& rowsPerPair, uint32_t world) {
    const size_t P = rowsPerPair.size();
    std::vector<std::pair<size_t, size_t>> out;
    if (world <= 1 || P == 0) {
        out.emplace_back(0, P);
        for (uint32_t r = 1; r < std::max(world, 1u); ++r) out.emplace_back(P, P);
        return out;
    }
    std::vector<uint64_t> csum(P + 1, 0);
    for (size_t i = 0; i < P; ++i) csum[i + 1] = csum[i] + std::max<uint64_t>(rowsPerPair[i], 1);
    std::vector<size_t> cuts(1, 0);
    for (uint32_t r = 1; r < world; ++r) {
        const double target = (double)(csum[P] * r) / (double)world;
        size_t k = 0;
        while (k <= P && (double)csum[k] < target) ++k;  // first prefix sum >= target
        cuts.push_back(std::min(std::max(k, cuts.back()), P));
    }
    cuts.push_back(P);
    for (uint32_t r = 0; r < world; ++r) out.emplace_back(cuts[r], cuts[r + 1]);
    return out;
}

}  // namespace dist
}  // namespace reconstruction
