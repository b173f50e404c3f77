// distributed.cpp -- implementation of host/distributed.hpp (part of libpgi_host.so).
#include "distributed.hpp"

#include <arpa/inet.h>
#include <hip/hip_runtime.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace reconstruction {
namespace dist {

namespace {
const char* envOr(const char* name, const char* fallback) {
    const char* v = std::getenv(name);
    return (v && *v) ? v : fallback;
}
}  // namespace

int selectDevice(const LaunchEnv& env) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw PgiError("selectDevice: no HIP device");
    const int dev = (int)(env.localRank % (uint32_t)n);
    if (hipSetDevice(dev) != hipSuccess) throw PgiError("selectDevice: hipSetDevice failed");
    return dev;
}

Transport attach(Engine& engine, HostComm& comm, Transport want) {
    if (comm.world() <= 1) return Transport::Host;  // nothing to attach: pgi_allgather_edges is a device copy
    if (want == Transport::Auto) {
        const std::string e = envOr("PGI_COMM", "auto");
        if (e == "host") want = Transport::Host;
        else if (e == "rccl") want = Transport::Rccl;
    }
    if (want == Transport::Auto) {
        // RCCL needs one device per rank: compare (host name, PCI bus id) across ranks
        struct Id { char s[96]; } mine{};
        int dev = 0;
        (void)hipGetDevice(&dev);
        char bus[32] = "?";
        (void)hipDeviceGetPCIBusId(bus, sizeof bus, dev);
        char host[48] = "?";
        (void)gethostname(host, sizeof host - 1);
        std::snprintf(mine.s, sizeof mine.s, "%s/%s", host, bus);
        const std::vector<Id> all = comm.allgather(mine);
        bool distinct = true;
        for (size_t i = 0; i < all.size(); ++i)
            for (size_t j = i + 1; j < all.size(); ++j) distinct &= std::strcmp(all[i].s, all[j].s) != 0;
        want = distinct ? Transport::Rccl : Transport::Host;
    }
    if (want == Transport::Rccl) {
        // ncclCommInitRank is a collective: a rank that fails BEFORE it (librccl missing, a broken device) would leave the
        // others blocked inside it for ever.  So every step that can fail on one rank only is followed by an agreement
        // round over the host star, and the ranks either all hold a communicator or all throw.
        auto agree = [&](bool mine, const char* what) {
            const std::vector<uint8_t> all = comm.allgather<uint8_t>(mine ? 1 : 0);
            std::string bad;
            for (size_t r = 0; r < all.size(); ++r)
                if (!all[r]) bad += (bad.empty() ? "" : ",") + std::to_string(r);
            if (!bad.empty())
                throw PgiError(std::string("attach: ") + what + " failed on rank(s) " + bad + (mine ? "" : std::string(": ") + pgi_last_error()));
        };
        int dev = 0;
        const bool usable = pgi_comm_rccl_probe(nullptr) == PGI_SUCCESS && hipGetDevice(&dev) == hipSuccess;
        agree(usable, "opening RCCL");
        uint8_t id[PGI_COMM_ID_BYTES] = {0};
        int rc = 0;
        if (comm.rank() == 0) rc = pgi_comm_unique_id(id);
        int32_t ok = rc == PGI_SUCCESS ? 1 : 0;
        comm.broadcast(&ok, 4);
        if (!ok) throw PgiError(std::string("attach: ") + (comm.rank() == 0 ? pgi_last_error() : "rank 0 could not create the RCCL id"));
        comm.broadcast(id, sizeof id);
        rc = pgi_comm_init_rccl(engine.get(), comm.world(), comm.rank(), id);
        try {
            agree(rc == PGI_SUCCESS, "pgi_comm_init_rccl");
        } catch (...) {
            if (rc == PGI_SUCCESS) (void)pgi_comm_destroy(engine.get());
            throw;
        }
        return Transport::Rccl;
    }
    Engine::check(pgi_comm_init_host(engine.get(), comm.world(), comm.rank(), &HostComm::transportCallback, &comm));
    return Transport::Host;
}

}  // namespace dist
}  // namespace reconstruction
