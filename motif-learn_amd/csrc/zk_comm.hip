// zk_comm.hip -- the one exchange step of the multi-GPU path: in-place all-gather of row blocks on RCCL
// (include/zernike_hip.h, "Multi-GPU").  One process per GPU; a communicator owns one ncclComm_t and one
// HIP stream; collectives are ordered after the producer's stream and run on the communicator's stream.
//
// RCCL is bound at run time (dlopen / dlsym) so that libzernike_hip.so loads on machines without it and
// never pulls a second copy of RCCL into a process that already has one (PyTorch-ROCm ships its own).
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <rccl/rccl.h>  // types and prototypes only; nothing here links against librccl

#include <new>

#include "zk_internal.h"

static_assert(ZK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

namespace {

struct rccl_api {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
};

rccl_api g_rccl;

// librccl.so.1 already mapped (e.g. by torch) -> the copy next to the HIP runtime this process uses -> the
// loader's default search -> /opt/rocm/lib
int rccl_load() {
  if (g_rccl.handle) return 0;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  if (!h) {
    Dl_info info;
    if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir(info.dli_fname);
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) {
        dir.resize(slash + 1);
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
          h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_GLOBAL);
          if (h) break;
        }
      }
    }
  }
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return zk_fail(ZK_E_COMM, std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "?"));
  rccl_api a;
  a.handle = h;
#define ZK_SYM(field, name)                                                                 \
  a.field = (decltype(a.field))dlsym(h, name);                                              \
  if (!a.field) return zk_fail(ZK_E_COMM, std::string("librccl lacks the symbol ") + name)
  ZK_SYM(GetUniqueId, "ncclGetUniqueId");
  ZK_SYM(CommInitRank, "ncclCommInitRank");
  ZK_SYM(CommDestroy, "ncclCommDestroy");
  ZK_SYM(GetErrorString, "ncclGetErrorString");
  ZK_SYM(AllGather, "ncclAllGather");
  ZK_SYM(Broadcast, "ncclBroadcast");
  ZK_SYM(Send, "ncclSend");
  ZK_SYM(Recv, "ncclRecv");
  ZK_SYM(GroupStart, "ncclGroupStart");
  ZK_SYM(GroupEnd, "ncclGroupEnd");
#undef ZK_SYM
  g_rccl = a;
  return 0;
}

int rccl_fail(ncclResult_t r, const char* what) {
  return zk_fail(ZK_E_COMM, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}

#define ZK_NCCL(call)                                    \
  do {                                                   \
    ncclResult_t zk_r_ = (call);                         \
    if (zk_r_ != ncclSuccess) return rccl_fail(zk_r_, #call); \
  } while (0)

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void nap_ms(int ms) {
  timespec ts = {ms / 1000, (long)(ms % 1000) * 1000000L};
  nanosleep(&ts, nullptr);
}

int send_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) {
    const ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return -1;
    }
    p += k;
    n -= (size_t)k;
  }
  return 0;
}

int recv_all(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n) {
    const ssize_t k = recv(fd, p, n, 0);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return -1;
    }
    p += k;
    n -= (size_t)k;
  }
  return 0;
}

}  // namespace

struct zk_comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;   // collectives run here
  hipEvent_t ev_in = nullptr;     // producer stream -> comm stream
  hipEvent_t ev_out = nullptr;    // comm stream -> consumer stream
  void* d_small = nullptr;        // zk_comm_allgather_host staging: (world + 1) * small_unit bytes, grows on demand
  size_t small_unit = 256;
  int algo = 0;                   // 0 auto, 1 p2p, 2 allgather, 3 bcast
};

extern "C" int zk_comm_rank(const zk_comm* c) { return c ? c->rank : -1; }
extern "C" int zk_comm_world(const zk_comm* c) { return c ? c->world : 0; }

extern "C" int zk_comm_unique_id(void* id_out) {
  if (!id_out) return zk_fail(ZK_E_BADARG, "id_out is null");
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  ZK_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, sizeof id);
  return 0;
}

extern "C" int zk_comm_destroy(zk_comm* c) {
  if (!c) return 0;
  zk_device_scope scope(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->nccl && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->d_small) (void)hipFree(c->d_small);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int zk_comm_init_rank(int device, int rank, int world, const void* id, zk_comm** out) {
  if (!out) return zk_fail(ZK_E_BADARG, "out is null");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world || !id) return zk_fail(ZK_E_BADARG, "need 0 <= rank < world and an id");
  int rc = rccl_load();
  if (rc) return rc;
  int ndev = 0;
  ZK_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return zk_fail(ZK_E_BADARG, "device index out of range");
  ZK_ON_DEVICE(device);
  zk_comm* c = new (std::nothrow) zk_comm();
  if (!c) return zk_fail(ZK_E_NOMEM, "out of host memory");
  c->device = device;
  c->rank = rank;
  c->world = world;
  if (const char* a = getenv("ZK_COMM_ALGO")) {
    if (!strcmp(a, "p2p")) c->algo = 1;
    else if (!strcmp(a, "allgather")) c->algo = 2;
    else if (!strcmp(a, "bcast")) c->algo = 3;
  }
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(&c->d_small, (size_t)(world + 1) * 256);
  if (e != hipSuccess) {
    rc = zk_hip_fail(e, "communicator stream / events");
    zk_comm_destroy(c);
    return rc;
  }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, uid, rank);
  if (r != ncclSuccess) {
    c->nccl = nullptr;
    rc = rccl_fail(r, "ncclCommInitRank");
    zk_comm_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

// ---- rendezvous through a file (ranks of one node) ---------------------------------------------------
extern "C" int zk_comm_init_file(int device, int rank, int world, const char* path, double timeout_s, zk_comm** out) {
  if (!out) return zk_fail(ZK_E_BADARG, "out is null");
  *out = nullptr;
  if (!path || !*path) return zk_fail(ZK_E_BADARG, "path is empty");
  if (world < 1 || rank < 0 || rank >= world) return zk_fail(ZK_E_BADARG, "need 0 <= rank < world");
  if (timeout_s <= 0) timeout_s = 120.0;
  char id[ZK_COMM_ID_BYTES];
  const std::string p(path);
  if (rank == 0) {
    // a file a dead earlier run left behind must not be taken for this run's id: remove it before anything else.  (A peer
    // can still read it in the instant before this line runs; callers close that window by naming the file per run, as
    // bench.py does with the launcher's pid and port.)
    (void)unlink(p.c_str());
    int rc = zk_comm_unique_id(id);
    if (rc) return rc;
    const std::string tmp = p + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return zk_fail(ZK_E_COMM, "cannot write " + tmp + ": " + strerror(errno));
    const bool ok = fwrite(id, 1, sizeof id, f) == sizeof id;
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), p.c_str()) != 0) {
      (void)unlink(tmp.c_str());
      return zk_fail(ZK_E_COMM, "cannot publish " + p + ": " + strerror(errno));
    }
  } else {
    const double t_end = now_s() + timeout_s;
    // rank 0 may have published up to timeout_s before this rank arrived (it waits in ncclCommInitRank); anything older
    // is a leftover of another run
    const time_t oldest = time(nullptr) - (time_t)timeout_s - 2;
    for (;;) {
      FILE* f = fopen(p.c_str(), "rb");
      if (f) {
        struct stat st;
        const bool fresh = fstat(fileno(f), &st) == 0 && st.st_mtime >= oldest;
        const size_t k = fresh ? fread(id, 1, sizeof id, f) : 0;
        fclose(f);
        if (k == sizeof id) break;
      }
      if (now_s() > t_end) return zk_fail(ZK_E_COMM, "timed out waiting for the id file " + p);
      nap_ms(20);
    }
  }
  const int rc = zk_comm_init_rank(device, rank, world, id, out);
  // ncclCommInitRank returns once every rank has joined: the file has served its purpose
  if (rank == 0) (void)unlink(p.c_str());
  return rc;
}

// ---- rendezvous over TCP: rank 0 listens, every peer connects, says its rank and receives the id ---------
extern "C" int zk_comm_init_tcp(int device, int rank, int world, const char* host, int port, double timeout_s,
                                zk_comm** out) {
  if (!out) return zk_fail(ZK_E_BADARG, "out is null");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return zk_fail(ZK_E_BADARG, "need 0 <= rank < world");
  if (port <= 0 || port > 65535) return zk_fail(ZK_E_BADARG, "bad port");
  if (!host || !*host) host = "127.0.0.1";
  if (timeout_s <= 0) timeout_s = 120.0;
  char id[ZK_COMM_ID_BYTES];
  const double t_end = now_s() + timeout_s;
  if (rank == 0) {
    int rc = zk_comm_unique_id(id);
    if (rc) return rc;
    if (world > 1) {
      const int ls = socket(AF_INET, SOCK_STREAM, 0);
      if (ls < 0) return zk_fail(ZK_E_COMM, std::string("socket: ") + strerror(errno));
      const int one = 1;
      setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
      sockaddr_in a = {};
      a.sin_family = AF_INET;
      a.sin_port = htons((uint16_t)port);
      // Listen on the address the peers were told to use rather than on every interface -- but only where that is known
      // to be an address of THIS machine the peers can reach: a literal IP keeps the narrow bind; a NAME that resolves to
      // loopback (the 127.0.1.1 many distributions map the local hostname to) would leave remote ranks timing out, and an
      // address that is not assigned locally (a VIP, a NAT or service address) cannot be bound at all -- both fall back
      // to every interface.  ZK_COMM_BIND_ADDR (an IPv4 literal, or "any") overrides.
      bool literal = inet_pton(AF_INET, host, &a.sin_addr) == 1;
      const char* force = getenv("ZK_COMM_BIND_ADDR");
      if (force && *force) {
        if (!strcmp(force, "any")) a.sin_addr.s_addr = htonl(INADDR_ANY);
        else if (inet_pton(AF_INET, force, &a.sin_addr) != 1) {
          close(ls);
          return zk_fail(ZK_E_COMM, std::string("ZK_COMM_BIND_ADDR is neither an IPv4 literal nor \"any\": ") + force);
        }
      } else if (!literal) {
        addrinfo hints = {}, *res = nullptr;
        hints.ai_family = AF_INET;
        hints.ai_socktype = SOCK_STREAM;
        if (getaddrinfo(host, nullptr, &hints, &res) != 0 || !res) {
          close(ls);
          return zk_fail(ZK_E_COMM, std::string("cannot resolve ") + host);
        }
        a.sin_addr = ((sockaddr_in*)res->ai_addr)->sin_addr;
        freeaddrinfo(res);
        const bool loopback = (ntohl(a.sin_addr.s_addr) >> 24) == 127;
        if (loopback && strcmp(host, "localhost") != 0) a.sin_addr.s_addr = htonl(INADDR_ANY);
      }
      int brc = bind(ls, (sockaddr*)&a, sizeof a);
      if (brc != 0 && errno == EADDRNOTAVAIL && !(force && *force)) {  // not an address of this machine: every interface
        a.sin_addr.s_addr = htonl(INADDR_ANY);
        brc = bind(ls, (sockaddr*)&a, sizeof a);
      }
      if (brc != 0 || listen(ls, world) != 0) {
        const std::string why = strerror(errno);
        close(ls);
        return zk_fail(ZK_E_COMM, "cannot listen on port " + std::to_string(port) + ": " + why);
      }
      std::vector<char> seen((size_t)world, 0);
      for (int got = 0; got < world - 1;) {
        timeval tv;
        const double left = t_end - now_s();
        if (left <= 0) {
          close(ls);
          return zk_fail(ZK_E_COMM, "timed out waiting for peers to connect");
        }
        tv.tv_sec = (long)left;
        tv.tv_usec = (long)((left - (double)tv.tv_sec) * 1e6);
        fd_set fds;
        FD_ZERO(&fds);
        FD_SET(ls, &fds);
        if (select(ls + 1, &fds, nullptr, nullptr, &tv) <= 0) continue;
        const int fd = accept(ls, nullptr, nullptr);
        if (fd < 0) continue;
        int32_t peer = -1;
        if (recv_all(fd, &peer, sizeof peer) == 0 && peer > 0 && peer < world && !seen[(size_t)peer] &&
            send_all(fd, id, sizeof id) == 0) {
          seen[(size_t)peer] = 1;
          ++got;
        }
        close(fd);
      }
      close(ls);
    }
  } else {
    addrinfo hints = {}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string ps = std::to_string(port);
    if (getaddrinfo(host, ps.c_str(), &hints, &res) != 0 || !res)
      return zk_fail(ZK_E_COMM, std::string("cannot resolve ") + host);
    bool done = false;
    while (!done) {
      const int fd = socket(AF_INET, SOCK_STREAM, 0);
      if (fd >= 0 && connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
        const int32_t me = rank;
        done = send_all(fd, &me, sizeof me) == 0 && recv_all(fd, id, sizeof id) == 0;
      }
      if (fd >= 0) close(fd);
      if (!done) {
        if (now_s() > t_end) {
          freeaddrinfo(res);
          return zk_fail(ZK_E_COMM, std::string("timed out connecting to ") + host + ":" + ps);
        }
        nap_ms(50);
      }
    }
    freeaddrinfo(res);
  }
  return zk_comm_init_rank(device, rank, world, id, out);
}

// ---- the collective ------------------------------------------------------------------------------------
namespace {

// rows [lo, hi) of rank r's block that the window [row_off, row_off + n_rows) selects
inline void window_of(int r, int64_t H, int64_t rpr, int64_t row_off, int64_t n_rows, int64_t* lo, int64_t* hi) {
  const int64_t b0 = (int64_t)r * rpr;
  int64_t b1 = b0 + rpr;
  if (b1 > H) b1 = H;
  int64_t a = b0 + row_off, z = b0 + row_off + n_rows;
  if (z > b1) z = b1;
  if (a > z) a = z;
  *lo = a;
  *hi = z;
}

int check_rows_args(int rank, int world, int64_t n_planes, int64_t H, int64_t W, int64_t rpr, int64_t row_off,
                    int64_t n_rows) {
  if (world < 1 || rank < 0 || rank >= world) return zk_fail(ZK_E_BADARG, "need 0 <= rank < world");
  if (n_planes < 0 || H < 0 || W < 0 || rpr < 0 || row_off < 0 || n_rows < 0)
    return zk_fail(ZK_E_BADARG, "negative extent");
  if (rpr * (int64_t)world < H) return zk_fail(ZK_E_BADARG, "rows_per_rank * world does not cover the rows");
  if (row_off + n_rows > rpr) return zk_fail(ZK_E_BADARG, "window exceeds the block");
  return 0;
}

// The schedule.  `emit(op, peer, group, plane, offset, count)` is called once per RCCL call, in issue order.
template <class Emit>
void plan_rows(int rank, int world, int64_t n_planes, int64_t H, int64_t W, int64_t rpr, int64_t row_off, int64_t n_rows,
               int algo, Emit&& emit) {
  if (world == 1 || n_planes == 0 || H == 0 || W == 0 || n_rows == 0) return;
  const int64_t plane = H * W;
  const bool whole_equal = n_planes == 1 && row_off == 0 && n_rows == rpr && rpr * (int64_t)world == H;
  if (algo == ZK_COMM_AUTO) algo = whole_equal ? ZK_COMM_ALLGATHER : ZK_COMM_P2P;
  if (algo == ZK_COMM_ALLGATHER && !whole_equal) algo = ZK_COMM_P2P;
  if (algo == ZK_COMM_ALLGATHER) {
    emit(ZK_XFER_ALLGATHER, -1, 0, 0, (int64_t)rank * rpr * W, rpr * W);
    return;
  }
  int64_t my_lo, my_hi;
  window_of(rank, H, rpr, row_off, n_rows, &my_lo, &my_hi);
  for (int64_t j = 0; j < n_planes; ++j) {
    // at most ZK_COMM_PLANES_PER_GROUP planes (that many sends + receives per peer) per group launch
    const int group = (int)(j / ZK_COMM_PLANES_PER_GROUP);
    const int64_t base = j * plane;
    if (algo == ZK_COMM_BCAST) {
      for (int owner = 0; owner < world; ++owner) {
        int64_t lo, hi;
        window_of(owner, H, rpr, row_off, n_rows, &lo, &hi);
        if (hi > lo) emit(ZK_XFER_BCAST, owner, group, (int)j, base + lo * W, (hi - lo) * W);
      }
      continue;
    }
    for (int step = 1; step < world; ++step) {
      // staggered peer order: at step s every rank sends to rank + s and receives from rank - s
      const int to = (rank + step) % world, from = (rank - step + world) % world;
      if (my_hi > my_lo) emit(ZK_XFER_SEND, to, group, (int)j, base + my_lo * W, (my_hi - my_lo) * W);
      int64_t lo, hi;
      window_of(from, H, rpr, row_off, n_rows, &lo, &hi);
      if (hi > lo) emit(ZK_XFER_RECV, from, group, (int)j, base + lo * W, (hi - lo) * W);
    }
  }
}

}  // namespace

extern "C" int zk_allgather_rows_plan(int rank, int world, int64_t n_planes, int64_t H, int64_t W, int64_t rpr,
                                      int64_t row_off, int64_t n_rows, int algo, zk_xfer* out, int64_t cap, int64_t* n_out) {
  if (!n_out) return zk_fail(ZK_E_BADARG, "n_out is null");
  *n_out = 0;
  if (algo < ZK_COMM_AUTO || algo > ZK_COMM_BCAST) return zk_fail(ZK_E_BADARG, "unknown algo");
  if (cap < 0 || (cap > 0 && !out)) return zk_fail(ZK_E_BADARG, "bad output list");
  const int rc = check_rows_args(rank, world, n_planes, H, W, rpr, row_off, n_rows);
  if (rc) return rc;
  int64_t n = 0;
  plan_rows(rank, world, n_planes, H, W, rpr, row_off, n_rows, algo,
            [&](int op, int peer, int group, int plane, int64_t offset, int64_t count) {
              if (n < cap) out[n] = zk_xfer{op, peer, group, plane, offset, count};
              ++n;
            });
  *n_out = n;
  return 0;
}

// The executor: the planner's list, in order, handed to RCCL on the communicator's stream.
extern "C" int zk_allgather_rows(zk_comm* c, double* full, int64_t n_planes, int64_t H, int64_t W, int64_t rpr,
                                 int64_t row_off, int64_t n_rows, void* hip_stream) {
  if (!c) return zk_fail(ZK_E_BADARG, "null communicator");
  int rc = check_rows_args(c->rank, c->world, n_planes, H, W, rpr, row_off, n_rows);
  if (rc) return rc;
  if (n_planes == 0 || H == 0 || W == 0 || n_rows == 0) return 0;
  if (!full) return zk_fail(ZK_E_BADARG, "null device pointer");
  ZK_ON_DEVICE(c->device);
  hipStream_t producer = (hipStream_t)hip_stream;
  ZK_HIP(hipEventRecord(c->ev_in, producer));
  ZK_HIP(hipStreamWaitEvent(c->stream, c->ev_in, 0));
  if (c->world == 1) return 0;
  ncclResult_t r = ncclSuccess;
  int open_group = -1;  // the group bracket currently open (collective-only plans of one entry need none)
  const char* failed = "";
  plan_rows(c->rank, c->world, n_planes, H, W, rpr, row_off, n_rows, c->algo,
            [&](int op, int peer, int group, int, int64_t offset, int64_t count) {
              if (r != ncclSuccess) return;
              if (op != ZK_XFER_ALLGATHER && group != open_group) {
                if (open_group >= 0) r = g_rccl.GroupEnd();
                if (r == ncclSuccess) r = g_rccl.GroupStart();
                if (r != ncclSuccess) {
                  failed = "grouped exchange (plane batch)";
                  return;
                }
                open_group = group;
              }
              double* at = full + offset;
              switch (op) {
                case ZK_XFER_SEND: r = g_rccl.Send(at, (size_t)count, ncclFloat64, peer, c->nccl, c->stream); break;
                case ZK_XFER_RECV: r = g_rccl.Recv(at, (size_t)count, ncclFloat64, peer, c->nccl, c->stream); break;
                case ZK_XFER_BCAST: r = g_rccl.Broadcast(at, at, (size_t)count, ncclFloat64, peer, c->nccl, c->stream); break;
                default: r = g_rccl.AllGather(at, at - (int64_t)c->rank * count, (size_t)count, ncclFloat64, c->nccl, c->stream);
              }
              if (r != ncclSuccess) failed = "grouped exchange";
            });
  if (open_group >= 0) {  // always close the bracket, also after a failed call inside it
    const ncclResult_t r_end = g_rccl.GroupEnd();
    if (r == ncclSuccess && r_end != ncclSuccess) {
      r = r_end;
      failed = "ncclGroupEnd";
    }
  }
  if (r != ncclSuccess) return rccl_fail(r, failed);
  return 0;
}

extern "C" int zk_comm_join(zk_comm* c, void* hip_stream) {
  if (!c) return zk_fail(ZK_E_BADARG, "null communicator");
  ZK_ON_DEVICE(c->device);
  ZK_HIP(hipEventRecord(c->ev_out, c->stream));
  ZK_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_out, 0));
  return 0;
}

extern "C" int zk_comm_allgather_host(zk_comm* c, const void* send_host, void* recv_host, int64_t bytes) {
  if (!c) return zk_fail(ZK_E_BADARG, "null communicator");
  if (bytes <= 0 || bytes > (1 << 26) || !send_host || !recv_host) return zk_fail(ZK_E_BADARG, "1 byte to 64 MiB per rank");
  ZK_ON_DEVICE(c->device);
  if ((size_t)bytes > c->small_unit) {  // every rank passes the same size, so every rank grows at the same call
    ZK_HIP(hipStreamSynchronize(c->stream));
    ZK_HIP(hipFree(c->d_small));
    c->d_small = nullptr;
    c->small_unit = ((size_t)bytes + 4095) & ~(size_t)4095;
    ZK_HIP(hipMalloc(&c->d_small, (size_t)(c->world + 1) * c->small_unit));
  }
  char* d = (char*)c->d_small;
  char* d_send = d + (size_t)c->world * c->small_unit;
  ZK_HIP(hipMemcpyAsync(d_send, send_host, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  if (c->world == 1) {
    ZK_HIP(hipMemcpyAsync(d, d_send, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
  } else {
    ZK_NCCL(g_rccl.AllGather(d_send, d, (size_t)bytes, ncclChar, c->nccl, c->stream));
  }
  ZK_HIP(hipMemcpyAsync(recv_host, d, (size_t)bytes * c->world, hipMemcpyDeviceToHost, c->stream));
  ZK_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- device memory helpers -------------------------------------------------------------------------------
extern "C" int zk_device_malloc(int device, int64_t bytes, void** out_dev) {
  if (!out_dev || bytes < 0) return zk_fail(ZK_E_BADARG, "bad arguments");
  *out_dev = nullptr;
  if (bytes == 0) return 0;
  ZK_ON_DEVICE(device);
  ZK_HIP(hipMalloc(out_dev, (size_t)bytes));
  return 0;
}

extern "C" int zk_device_free(int device, void* dev) {
  if (!dev) return 0;
  ZK_ON_DEVICE(device);
  ZK_HIP(hipFree(dev));
  return 0;
}

extern "C" int zk_device_copy(int device, void* dst, const void* src, int64_t bytes, int kind) {
  if (bytes < 0 || kind < 1 || kind > 3) return zk_fail(ZK_E_BADARG, "bad arguments");
  if (bytes == 0) return 0;
  if (!dst || !src) return zk_fail(ZK_E_BADARG, "null pointer");
  ZK_ON_DEVICE(device);
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  ZK_HIP(hipMemcpy(dst, src, (size_t)bytes, k));
  return 0;
}

// ---- plain-stream probes (see zernike_hip.h) ---------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(64) void probe_read_kernel(const char* __restrict__ in, long long n_groups, float* __restrict__ sink,
                                                        char* __restrict__ res, long long store_per_group) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int lane = threadIdx.x;
  float sum = 0.f;
  for (long long g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const char* base = in + g * 262144;
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + s * 16384 + i * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(lds + i * 256), 16, 0, 2 /* nt */);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      sum += lds[(lane * 7 + s) & 4095];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (res) {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2* dst = (d2*)(res + g * store_per_group);
      const d2 v = {(double)sum, 1.0};
      for (long long k = lane; k < store_per_group / 16; k += 64) __builtin_nontemporal_store(v, dst + k);
    }
  }
  if (sum == 1.2345e-30f) sink[blockIdx.x] = sum;  // keeps the reads alive; never true for real data
}

typedef float probe_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe_copy_kernel(const probe_f4* __restrict__ in, probe_f4* __restrict__ out, long long n16) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

}  // namespace

extern "C" int zk_hbm_probe(int device, const void* src, void* dst, int64_t bytes, int64_t store_per_group, int reps, double* ms_out) {
  if (!src || !ms_out || reps < 1 || bytes < 262144) return zk_fail(ZK_E_BADARG, "need a source, reps >= 1 and at least 256 KiB");
  if (store_per_group < 0 || store_per_group % 16 || (store_per_group && !dst)) return zk_fail(ZK_E_BADARG, "bad store size");
  ZK_ON_DEVICE(device);
  const long long n_groups = bytes / 262144;
  float* sink = nullptr;
  ZK_HIP(hipMalloc((void**)&sink, 8192 * sizeof(float)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  auto launch = [&]() {
    if (dst && !store_per_group)
      hipLaunchKernelGGL(probe_copy_kernel, dim3(8192), dim3(256), 0, 0, (const probe_f4*)src, (probe_f4*)dst, n_groups * 16384);
    else
      hipLaunchKernelGGL(probe_read_kernel, dim3(2048), dim3(64), 0, 0, (const char*)src, n_groups, sink, (char*)dst,
                         (long long)store_per_group);
  };
  float ms = 0.f;
  if (e == hipSuccess) {
    launch();
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipEventRecord(e0, 0);
  if (e == hipSuccess) {
    for (int k = 0; k < reps; ++k) launch();
    e = hipEventRecord(e1, 0);
  }
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (e != hipSuccess) return zk_hip_fail(e, "zk_hbm_probe");
  *ms_out = (double)ms / reps;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Shader clock while other kernels run (bench.py: every FP64-bound section is reported against the clock it actually ran
// at -- the chip's power management holds this class of kernel at 1.7-2.2 GHz of the nominal 2.4, profiles/r04_strip_trace.txt).
// One wave on a stream of its own: it reads the shader-clock counter (s_memtime) against the constant 100-MHz counter
// (s_memrealtime), then sleeps in short naps until the host clears its run flag (page-locked, mapped) or its budget of
// 100-MHz ticks is spent -- an exit every wave reaches whatever the host does -- and reads both again.
// ------------------------------------------------------------------------------------------------------------------------
struct zk_clock_monitor {
  int device = 0;
  hipStream_t stream = nullptr;
  volatile int* h_flag = nullptr;        // host-mapped: 1 = keep running
  unsigned long long* h_out = nullptr;   // host-mapped: rt0, ck0, rt1, ck1
};

namespace {
__global__ void clock_monitor_kernel(volatile int* flag, unsigned long long* out, unsigned long long budget_ticks, int mode) {
  if (threadIdx.x != 0 && mode != 2) return;
  const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), ck0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = rt0;
    out[1] = ck0;
  }
  unsigned long long rt = rt0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0;
  for (int spin = 0; spin < (1 << 24); ++spin) {             // (hard cap on iterations besides the tick budget)
    if (mode == 0) __builtin_amdgcn_s_sleep(100);            // ~6400 clocks: the wave costs its SIMD next to nothing
    if (mode == 2) {
#pragma unroll
      for (int k = 0; k < 64; ++k) b = __builtin_fma(b, a, 1e-300);
    }
    rt = __builtin_amdgcn_s_memrealtime();
    if (rt - rt0 >= budget_ticks || __hip_atomic_load((const int*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) break;
  }
  if (threadIdx.x == 0) {
    out[2] = __builtin_amdgcn_s_memrealtime();
    out[3] = __builtin_amdgcn_s_memtime();
    out[4] = (unsigned long long)b;
  }
}
}  // namespace

extern "C" int zk_clock_monitor_start(int device, double max_ms, zk_clock_monitor** out) {
  if (!out || !(max_ms > 0.0) || max_ms > 2000.0) return zk_fail(ZK_E_BADARG, "need out and 0 < max_ms <= 2000");
  *out = nullptr;
  ZK_ON_DEVICE(device);
  zk_clock_monitor* m = new (std::nothrow) zk_clock_monitor();
  if (!m) return zk_fail(ZK_E_NOMEM, "out of host memory");
  m->device = device;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipHostMalloc((void**)&m->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostMalloc((void**)&m->h_out, 64, hipHostMallocMapped | hipHostMallocCoherent);
  int* d_flag = nullptr;
  unsigned long long* d_out = nullptr;
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&d_flag, (void*)m->h_flag, 0);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&d_out, (void*)m->h_out, 0);
  if (e == hipSuccess) {
    *m->h_flag = 1;
    for (int k = 0; k < 4; ++k) m->h_out[k] = 0;
    const char* md = getenv("ZK_CLOCK_MONITOR_MODE");
    hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, m->stream, (volatile int*)d_flag, d_out,
                       (unsigned long long)(max_ms * 1e5), md ? atoi(md) : 0);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    if (m->stream) (void)hipStreamDestroy(m->stream);
    if (m->h_flag) (void)hipHostFree((void*)m->h_flag);
    if (m->h_out) (void)hipHostFree(m->h_out);
    delete m;
    return zk_hip_fail(e, "zk_clock_monitor_start");
  }
  *out = m;
  return 0;
}

extern "C" int zk_clock_monitor_stop(zk_clock_monitor* m, double* ghz_out, double* ms_out) {
  if (!m) return zk_fail(ZK_E_BADARG, "null monitor");
  ZK_ON_DEVICE(m->device);
  __atomic_store_n((int*)m->h_flag, 0, __ATOMIC_SEQ_CST);
  hipError_t e = hipStreamSynchronize(m->stream);
  const double ticks = (double)(m->h_out[2] - m->h_out[0]), clocks = (double)(m->h_out[3] - m->h_out[1]);
  if (ghz_out) *ghz_out = ticks > 0 ? clocks / (ticks * 10.0) : 0.0;
  if (ms_out) *ms_out = ticks / 1e5;
  (void)hipStreamDestroy(m->stream);
  (void)hipHostFree((void*)m->h_flag);
  (void)hipHostFree(m->h_out);
  delete m;
  if (e != hipSuccess) return zk_hip_fail(e, "zk_clock_monitor_stop");
  return 0;
}

extern "C" int zk_device_synchronize(int device) {
  ZK_ON_DEVICE(device);
  ZK_HIP(hipDeviceSynchronize());
  return 0;
}
