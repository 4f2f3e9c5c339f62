// sxfir_comm_*: the one exchange step of the sharded path -- the gather of the decimated channels to a root GPU
// over xGMI -- as plain C entry points over librccl, so that a C / C++ host (the thing that drives the kernels
// behind driver=sx) can run it without Python or torch.distributed.  SURVEY 8(e): channels are independent (the
// reference is single-channel, SoapySX.cpp:1591-1595), 64 channels -> 8 per GPU, and the gather is
//     ncclGroupStart; root: ncclRecv x (N-1); peers: ncclSend; ncclGroupEnd
// on the caller's HIP stream, in pieces, so that it queues behind the kernel that produced the block and the
// root can consume a step's first channels while the rest are on the links.
//
// librccl is loaded on first use (dlopen), not linked, and not needed at compile time either: the handful of RCCL
// prototypes used here are declared below (their ABI is NCCL's public one), so a single-GPU build of libsxfir.so
// needs neither the RCCL headers nor the library.  dlopen by SONAME hands back the copy the process already has
// (torch loads librccl too), so a process ends up with ONE RCCL, whoever asked first.
//
// Every entry point that has to make the communicator's GPU the calling thread's current device puts the previous
// one back before it returns (DeviceScope): the caller's HIP context is left as it was found.
//
// Included at the end of sxfir.hip (shares its error helper); not a stand-alone translation unit.
#pragma once

#include <dlfcn.h>

#include <mutex>

// NCCL's public C ABI, as far as this file uses it (nccl.h / rccl.h: ncclUniqueId is 128 opaque bytes, ncclComm_t an
// opaque pointer, ncclSuccess = 0, ncclChar = 0)
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { ncclSuccess = 0, ncclChar = 0 };
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommInitAll(ncclComm_t *, int ndev, const int *devlist);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommCount(const ncclComm_t, int *count);
ncclResult_t ncclCommUserRank(const ncclComm_t, int *rank);
ncclResult_t ncclCommCuDevice(const ncclComm_t, int *device);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void *, size_t count, ncclDataType_t, int peer, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void *, size_t count, ncclDataType_t, int peer, ncclComm_t, hipStream_t);
const char *ncclGetErrorString(ncclResult_t);
}

namespace {

// makes `device` current for the scope and restores what was current before
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
    }
    ~DeviceScope()
    {
        int now = -1;
        if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char why[256] = "";
};

Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl *rccl()
{
    std::call_once(g_rccl_once, [] {
        Rccl &r = g_rccl;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
            snprintf(r.why, sizeof(r.why), "%s", dlerror());
        }
        if (!r.handle) return;
#define SXFIR_SYM(NAME) \
    r.NAME = reinterpret_cast<decltype(r.NAME)>(dlsym(r.handle, "nccl" #NAME)); \
    if (!r.NAME) { snprintf(r.why, sizeof(r.why), "librccl lacks nccl" #NAME); dlclose(r.handle); r.handle = nullptr; return; }
        SXFIR_SYM(GetUniqueId) SXFIR_SYM(CommInitRank) SXFIR_SYM(CommInitAll) SXFIR_SYM(CommDestroy)
        SXFIR_SYM(CommCount) SXFIR_SYM(CommUserRank) SXFIR_SYM(CommCuDevice)
        SXFIR_SYM(GroupStart) SXFIR_SYM(GroupEnd) SXFIR_SYM(Send) SXFIR_SYM(Recv) SXFIR_SYM(GetErrorString)
#undef SXFIR_SYM
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

#define RCCLCHECK(R, expr)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(SXFIR_EHIP, "%s failed: %s", #expr, (R)->GetErrorString(r_));    \
    } while (0)

}  // namespace

struct sxfir_comm {
    ncclComm_t comm;
    int rank, nranks, device;
};

static_assert(sizeof(ncclUniqueId) == SXFIR_COMM_ID_BYTES, "SXFIR_COMM_ID_BYTES must be sizeof(ncclUniqueId)");

static int need_gpu_and_rccl(const Rccl **r)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(SXFIR_ENODEVICE, "no GPU visible: no collective without one");
    *r = rccl();
    if (!*r) return fail(SXFIR_EUNSUPPORTED, "librccl could not be loaded: %s", g_rccl.why);
    return SXFIR_OK;
}

int sxfir_comm_unique_id(void *id)
{
    if (!id) return fail(SXFIR_EINVAL, "id is NULL");
    const Rccl *R;
    if (int rc = need_gpu_and_rccl(&R)) return rc;
    ncclUniqueId u;
    RCCLCHECK(R, R->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SXFIR_OK;
}

int sxfir_comm_init_rank(sxfir_comm **out, const void *id, int nranks, int rank, int device)
{
    if (!out || !id) return fail(SXFIR_EINVAL, "NULL argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(SXFIR_EINVAL, "rank %d of %d", rank, nranks);
    const Rccl *R;
    if (int rc = need_gpu_and_rccl(&R)) return rc;
    if (device < 0) HIPCHECK(hipGetDevice(&device));
    DeviceScope on(device);
    HIPCHECK(on.err);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c;
    RCCLCHECK(R, R->CommInitRank(&c, nranks, u, rank));
    *out = new (std::nothrow) sxfir_comm{c, rank, nranks, device};
    if (!*out) { R->CommDestroy(c); return fail(SXFIR_ENOMEM, "out of memory"); }
    return SXFIR_OK;
}

int sxfir_comm_init_all(sxfir_comm **comms, int ndev, const int *devices)
{
    if (!comms || ndev < 1 || ndev > 64) return fail(SXFIR_EINVAL, "bad argument");
    for (int i = 0; i < ndev; ++i) comms[i] = nullptr;
    const Rccl *R;
    if (int rc = need_gpu_and_rccl(&R)) return rc;
    int devs[64];
    for (int i = 0; i < ndev; ++i) devs[i] = devices ? devices[i] : i;
    ncclComm_t c[64];
    int cur = 0;
    HIPCHECK(hipGetDevice(&cur));
    DeviceScope back(cur);                    // ncclCommInitAll visits every GPU: put the caller's back afterwards
    RCCLCHECK(R, R->CommInitAll(c, ndev, devs));
    for (int i = 0; i < ndev; ++i) comms[i] = new (std::nothrow) sxfir_comm{c[i], i, ndev, devs[i]};
    for (int i = 0; i < ndev; ++i) {
        if (comms[i]) continue;
        // all or nothing: no half-initialised set of communicators is handed back
        for (int k = 0; k < ndev; ++k) {
            R->CommDestroy(c[k]);
            delete comms[k];
            comms[k] = nullptr;
        }
        return fail(SXFIR_ENOMEM, "out of memory");
    }
    return SXFIR_OK;
}

int sxfir_comm_destroy(sxfir_comm *c)
{
    if (!c) return SXFIR_OK;
    const Rccl *R = rccl();
    if (R) {
        DeviceScope on(c->device);
        R->CommDestroy(c->comm);
    }
    delete c;
    return SXFIR_OK;
}

int sxfir_comm_rank(const sxfir_comm *c, int *rank, int *nranks, int *device)
{
    if (!c) return fail(SXFIR_EINVAL, "comm is NULL");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    if (device) *device = c->device;
    return SXFIR_OK;
}

int sxfir_comm_query(const sxfir_comm *c, int *rank, int *nranks, int *device)
{
    if (!c) return fail(SXFIR_EINVAL, "comm is NULL");
    const Rccl *R = rccl();
    if (!R) return fail(SXFIR_EUNSUPPORTED, "librccl is not loaded");
    int v = -1;
    if (nranks) { RCCLCHECK(R, R->CommCount(c->comm, &v)); *nranks = v; }
    if (rank) { RCCLCHECK(R, R->CommUserRank(c->comm, &v)); *rank = v; }
    if (device) { RCCLCHECK(R, R->CommCuDevice(c->comm, &v)); *device = v; }
    return SXFIR_OK;
}

// one rank's share of one piece [off, off + n) of the gather; the caller has opened the RCCL group
static int gather_piece(const Rccl *R, sxfir_comm *c, const char *send, char *recv, size_t off, size_t n, size_t stride,
                        int root, hipStream_t st)
{
    if (c->rank == root) {
        for (int r = 0; r < c->nranks; ++r)
            if (r != root) RCCLCHECK(R, R->Recv(recv + (size_t)r * stride + off, n, ncclChar, r, c->comm, st));
    } else {
        RCCLCHECK(R, R->Send(send + off, n, ncclChar, root, c->comm, st));
    }
    return SXFIR_OK;
}

static int gather_args(const sxfir_comm *c, const void *send, const void *recv, size_t bytes, size_t stride, int root)
{
    if (!c) return fail(SXFIR_EINVAL, "comm is NULL");
    if (root < 0 || root >= c->nranks) return fail(SXFIR_EINVAL, "root %d of %d ranks", root, c->nranks);
    if (bytes && !send) return fail(SXFIR_EINVAL, "send buffer is NULL");
    if (c->rank == root && bytes && (!recv || stride < bytes)) return fail(SXFIR_EINVAL, "root needs a receive buffer with stride >= bytes");
    return SXFIR_OK;
}

int sxfir_comm_gather(sxfir_comm *c, const void *send_dev, void *recv_dev, size_t bytes, size_t recv_stride_bytes, int root,
                      size_t chunk_bytes, void *stream)
{
    if (int rc = gather_args(c, send_dev, recv_dev, bytes, recv_stride_bytes, root)) return rc;
    if (bytes == 0) return SXFIR_OK;
    const Rccl *R = rccl();
    if (!R) return fail(SXFIR_EUNSUPPORTED, "librccl is not loaded");
    DeviceScope on(c->device);
    HIPCHECK(on.err);
    const char *send = static_cast<const char *>(send_dev);
    char *recv = static_cast<char *>(recv_dev);
    const size_t piece = chunk_bytes ? chunk_bytes : bytes;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = std::min(piece, bytes - off);
        if (c->nranks > 1) {
            RCCLCHECK(R, R->GroupStart());
            const int rc = gather_piece(R, c, send, recv, off, n, recv_stride_bytes, root, S(stream));
            RCCLCHECK(R, R->GroupEnd());
            if (rc) return rc;
        }
        // the root's own block: in place already, or one device-to-device copy on the same stream
        if (c->rank == root && recv + (size_t)root * recv_stride_bytes != send)
            HIPCHECK(hipMemcpyAsync(recv + (size_t)root * recv_stride_bytes + off, send + off, n, hipMemcpyDeviceToDevice, S(stream)));
    }
    return SXFIR_OK;
}

int sxfir_comm_gather_all(sxfir_comm *const *comms, int ndev, const void *const *send_dev, void *recv_dev, size_t bytes,
                          size_t recv_stride_bytes, int root, size_t chunk_bytes, void *const *streams)
{
    if (!comms || !send_dev || ndev < 1) return fail(SXFIR_EINVAL, "bad argument");
    for (int i = 0; i < ndev; ++i) {
        if (!comms[i] || comms[i]->nranks != ndev || comms[i]->rank != i) return fail(SXFIR_EINVAL, "comms[%d] is not rank %d of %d", i, i, ndev);
        if (int rc = gather_args(comms[i], send_dev[i], recv_dev, bytes, recv_stride_bytes, root)) return rc;
    }
    if (bytes == 0) return SXFIR_OK;
    const Rccl *R = rccl();
    if (!R) return fail(SXFIR_EUNSUPPORTED, "librccl is not loaded");
    char *recv = static_cast<char *>(recv_dev);
    const size_t piece = chunk_bytes ? chunk_bytes : bytes;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = std::min(piece, bytes - off);
        if (ndev > 1) {
            // one thread drives every rank: all of a piece's sends and receives inside ONE group
            RCCLCHECK(R, R->GroupStart());
            int rc = SXFIR_OK;
            for (int i = 0; i < ndev && !rc; ++i)
                rc = gather_piece(R, comms[i], static_cast<const char *>(send_dev[i]), recv, off, n, recv_stride_bytes, root,
                                  S(streams ? streams[i] : nullptr));
            RCCLCHECK(R, R->GroupEnd());
            if (rc) return rc;
        }
        const char *own = static_cast<const char *>(send_dev[root]);
        if (recv + (size_t)root * recv_stride_bytes != own) {
            DeviceScope on(comms[root]->device);
            HIPCHECK(on.err);
            HIPCHECK(hipMemcpyAsync(recv + (size_t)root * recv_stride_bytes + off, own + off, n, hipMemcpyDeviceToDevice,
                                    S(streams ? streams[root] : nullptr)));
        }
    }
    return SXFIR_OK;
}
