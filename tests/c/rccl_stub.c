/* rccl_stub.c -- TEST INFRASTRUCTURE ONLY: the four RCCL entry points libpmc_hip.so uses (ncclGetUniqueId, ncclCommInitRank,
 * ncclAllGather, ncclCommDestroy; particlesmc_amd/csrc/pmc_hip.hip "RcclApi"), implemented over files in a directory, so that
 * two or more PROCESSES that share ONE GPU can drive pmc_rccl_comm_create + pmc_replica_exchange_rccl end to end -- the id file
 * hand-over, the rendezvous, the ordering of the collective on the handle's stream, the label broadcast -- on a box with a
 * single device (the real RCCL refuses two ranks on one device).  Loaded instead of librccl.so through PMC_RCCL_LIB.  Nothing
 * in the product links or ships this; tests/test_gpu_zz_multirank.py builds it with hipcc on the GPU box.
 *
 * Semantics kept from RCCL: ncclCommInitRank returns only after every rank has joined; ncclAllGather is ordered after the work
 * already enqueued on `stream` and its result is visible to work enqueued after it (here: it synchronises the stream, which is
 * stricter); every rank receives the ranks' contributions in rank order.  */
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7,
               ncclFloat64 = 8 } ncclDataType_t;
typedef struct stub_comm {
    char dir[128];
    int rank, nranks;
    unsigned long seq;
} *ncclComm_t;

static void nap(void) { struct timespec t = {0, 2000000}; nanosleep(&t, NULL); }

static int wait_for(const char *path, double timeout_s)
{
    const time_t t0 = time(NULL);
    struct stat st;
    while (stat(path, &st) != 0) {
        if (difftime(time(NULL), t0) > timeout_s) return -1;
        nap();
    }
    return 0;
}

static int put(const char *dir, const char *name, const void *data, size_t n)
{
    char tmp[320], fin[320];
    snprintf(tmp, sizeof tmp, "%s/%s.tmp%d", dir, name, (int)getpid());
    snprintf(fin, sizeof fin, "%s/%s", dir, name);
    FILE *f = fopen(tmp, "wb");
    if (!f) return -1;
    if (n && fwrite(data, 1, n, f) != n) { fclose(f); return -1; }
    fclose(f);
    return rename(tmp, fin);
}

static int get(const char *dir, const char *name, void *data, size_t n, double timeout_s)
{
    char fin[320];
    snprintf(fin, sizeof fin, "%s/%s", dir, name);
    if (wait_for(fin, timeout_s)) return -1;
    FILE *f = fopen(fin, "rb");
    if (!f) return -1;
    const size_t got = n ? fread(data, 1, n, f) : 0;
    fclose(f);
    return got == n ? 0 : -1;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "rccl_stub: rendezvous or copy failed"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id->internal, 0, sizeof id->internal);
    const char *base = getenv("TMPDIR");
    snprintf(id->internal, sizeof id->internal, "%s/pmc_rccl_stub_%d_%ld", base ? base : "/tmp", (int)getpid(), (long)time(NULL));
    if (mkdir(id->internal, 0700) != 0 && errno != EEXIST) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    struct stub_comm *c = (struct stub_comm *)calloc(1, sizeof *c);
    memcpy(c->dir, id.internal, sizeof c->dir);
    c->dir[sizeof c->dir - 1] = 0;
    c->rank = rank;
    c->nranks = nranks;
    char name[64];
    snprintf(name, sizeof name, "join_%d", rank);
    if (put(c->dir, name, "", 0)) { free(c); return ncclSystemError; }
    for (int r = 0; r < nranks; r++) {                  /* collective: every rank has joined before anyone returns */
        snprintf(name, sizeof name, "join_%d", r);
        if (get(c->dir, name, NULL, 0, 120.0)) { free(c); return ncclSystemError; }
    }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    const size_t es = (datatype == ncclFloat64 || datatype == ncclInt64 || datatype == ncclUint64) ? 8 :
                      (datatype == ncclInt8 || datatype == ncclUint8) ? 1 : (datatype == ncclFloat16) ? 2 : 4;
    const size_t nb = sendcount * es;
    char *host = (char *)malloc(nb * (size_t)comm->nranks);
    hipError_t he = hipStreamSynchronize(stream);                                              /* ordered after the stream's work */
    if (he != hipSuccess) { fprintf(stderr, "rccl_stub[%d]: stream sync: %s\n", comm->rank, hipGetErrorString(he)); free(host); return ncclUnhandledCudaError; }
    he = hipMemcpy(host + nb * (size_t)comm->rank, sendbuff, nb, hipMemcpyDeviceToHost);
    if (he != hipSuccess) { fprintf(stderr, "rccl_stub[%d]: D2H: %s\n", comm->rank, hipGetErrorString(he)); free(host); return ncclUnhandledCudaError; }
    char name[64];
    snprintf(name, sizeof name, "ag_%lu_%d", comm->seq, comm->rank);
    if (put(comm->dir, name, host + nb * (size_t)comm->rank, nb)) { fprintf(stderr, "rccl_stub[%d]: put %s/%s: %s\n", comm->rank, comm->dir, name, strerror(errno)); free(host); return ncclSystemError; }
    for (int r = 0; r < comm->nranks; r++) {
        if (r == comm->rank) continue;
        snprintf(name, sizeof name, "ag_%lu_%d", comm->seq, r);
        if (get(comm->dir, name, host + nb * (size_t)r, nb, 60.0)) { fprintf(stderr, "rccl_stub[%d]: no %s/%s after 60 s (%s)\n", comm->rank, comm->dir, name, strerror(errno)); free(host); return ncclSystemError; }
    }
    if (comm->seq >= 2) {                               /* every rank has long read round seq - 2 */
        char old[320];
        snprintf(old, sizeof old, "%s/ag_%lu_%d", comm->dir, comm->seq - 2, comm->rank);
        (void)unlink(old);
    }
    comm->seq++;
    const hipError_t e = hipMemcpyAsync(recvbuff, host, nb * (size_t)comm->nranks, hipMemcpyHostToDevice, stream);
    const hipError_t e2 = hipStreamSynchronize(stream);
    free(host);
    return (e == hipSuccess && e2 == hipSuccess) ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclInvalidArgument;
    char p[320], name[64];
    /* a rank may only remove what it published once every rank has read it: leave together */
    snprintf(name, sizeof name, "leave_%d", comm->rank);
    (void)put(comm->dir, name, "", 0);
    for (int r = 0; r < comm->nranks; r++) {
        snprintf(name, sizeof name, "leave_%d", r);
        (void)get(comm->dir, name, NULL, 0, 60.0);
    }
    for (unsigned long s = comm->seq >= 2 ? comm->seq - 2 : 0; s < comm->seq; s++) {
        snprintf(p, sizeof p, "%s/ag_%lu_%d", comm->dir, s, comm->rank);
        (void)unlink(p);
    }
    /* (the empty join_ / leave_ markers stay: another rank may still be looking for them; the directory lives in the test's tmp) */
    free(comm);
    return ncclSuccess;
}
