// slab_rccl.cpp — the multi-GPU step loop in C++: one process per GPU, the slab exchange of include/nereus_hip.h
// (nrs_slab_pack -> one message per neighbour -> nrs_slab_unpack -> nrs_step) carried by RCCL send/recv on the context's own HIP
// stream, the IISPH exit test by one ncclAllReduce of the (sum, count) pair per solver iteration (sph_cuda.cu:736-741,816-819 made
// global).  It is what nereus_amd/slab.py's SlabDriver does through torch.distributed, without an interpreter on the critical path:
// a C++ caller of Nereus::SPH (main.cpp:533-616) that wants more than one GPU links this.
//
//   RANK=r WORLD_SIZE=w LOCAL_RANK=d NEREUS_NCCL_ID_FILE=/tmp/id  slab_rccl <rank scene.bin> <steps> <out.bin>
//
// scene.bin (written by tests / nereus_amd.slab.write_rank_scene): u32 magic 'NRSL', u32 n, u32 nb, i32 cell_lo, i32 cell_hi,
//   i32 halo, u32 msg_capacity, u32 ctx_capacity, u32 solver (0 SESPH / 1 IISPH), u32 sizeof(params); params; pos4[n]; vel4[n];
//   bi4[nb]; vbi[nb]  (fp32)
// out.bin: u32 n_owned, u32 truncated_steps, u32 last_iterations, u32 0; pos4[n_owned]; vel4[n_owned]
// The communicator is bootstrapped without MPI: rank 0 writes the ncclUniqueId to NEREUS_NCCL_ID_FILE (atomically, by rename), the
// other ranks wait for the file.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include "nereus_hip.h"

static void die(const std::string &m) { std::fprintf(stderr, "slab_rccl: %s\n", m.c_str()); std::exit(2); }
#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) die(std::string(#e) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCL_OK(e) do { ncclResult_t r_ = (e); if (r_ != ncclSuccess) die(std::string(#e) + ": " + ncclGetErrorString(r_)); } while (0)
#define NRS_OK_(e) do { int r_ = (e); if (r_ != 0) die(std::string(#e) + ": " + nrs_last_error()); } while (0)

static int env_int(const char *k, int dflt) { const char *v = std::getenv(k); return v ? std::atoi(v) : dflt; }

static ncclUniqueId bootstrap_id(int rank)
{
    const char *path = std::getenv("NEREUS_NCCL_ID_FILE");
    if (!path) die("NEREUS_NCCL_ID_FILE is not set");
    ncclUniqueId id;
    if (rank == 0) {
        NCCL_OK(ncclGetUniqueId(&id));
        const std::string tmp = std::string(path) + ".tmp";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) die("cannot write the id file");
        std::fclose(f);
        if (std::rename(tmp.c_str(), path) != 0) die("cannot publish the id file");
        return id;
    }
    for (int tries = 0; tries < 6000; ++tries) { // up to a minute
        FILE *f = std::fopen(path, "rb");
        if (f) {
            const size_t got = std::fread(&id, sizeof(id), 1, f);
            std::fclose(f);
            if (got == 1) return id;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    die("timed out waiting for rank 0's id file");
    return id;
}

int main(int argc, char **argv)
{
    if (argc < 4) die("usage: slab_rccl <rank scene.bin> <steps> <out.bin>");
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), dev = env_int("LOCAL_RANK", 0);
    const int steps = std::atoi(argv[2]);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) die("cannot open the scene file");
    uint32_t hd[10];
    if (std::fread(hd, 4, 10, f) != 10 || hd[0] != 0x4c53524eu) die("bad scene header");
    const uint32_t n = hd[1], nb = hd[2], msgCap = hd[6], ctxCap = hd[7], solver = hd[8], psize = hd[9];
    const int32_t lo = (int32_t)hd[3], hi = (int32_t)hd[4], halo = (int32_t)hd[5];
    if (psize != sizeof(nrs_params_f32)) die("parameter block size mismatch");
    nrs_params_f32 P;
    std::vector<float> pos(4 * (size_t)n), vel(4 * (size_t)n), bi(4 * (size_t)nb), vbi(nb);
    if (std::fread(&P, psize, 1, f) != 1 || (n && (std::fread(pos.data(), 16, n, f) != n || std::fread(vel.data(), 16, n, f) != n)) ||
        (nb && (std::fread(bi.data(), 16, nb, f) != nb || std::fread(vbi.data(), 4, nb, f) != nb)))
        die("short scene file");
    std::fclose(f);

    HIP_OK(hipSetDevice(dev));
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); // ONE stream for the solver and the exchange: stream order is the only
                                                                     // ordering between pack, send/recv and unpack
    ncclComm_t comm;
    const ncclUniqueId id = bootstrap_id(rank);
    NCCL_OK(ncclCommInitRank(&comm, world, id, rank));

    nrs_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg); cfg.device = dev; cfg.solver = (int32_t)solver; cfg.precision = 32; cfg.kernel_set = NRS_KERNELS_MULLER;
    cfg.surface_tension = 1; cfg.flags = solver == NRS_SOLVER_IISPH ? NRS_FLAG_IISPH_SELF_BY_SLOT : 0; cfg.capacity = ctxCap; cfg.stream = stream;
    nrs_ctx *ctx = nullptr;
    NRS_OK_(nrs_create(&cfg, &P, &ctx));
    NRS_OK_(nrs_upload_particles(ctx, pos.data(), vel.data(), nullptr, 0, n));
    NRS_OK_(nrs_set_boundaries(ctx, nb ? bi.data() : nullptr, nb ? vbi.data() : nullptr, nb, 0));
    NRS_OK_(nrs_slab_configure(ctx, lo, hi, halo));

    const int left = rank > 0 ? rank - 1 : -1, right = rank < world - 1 ? rank + 1 : -1;
    const size_t msgBytes = (size_t)nrs_slab_message_bytes(msgCap, 32);
    void *sendL = nullptr, *sendR = nullptr, *recvL = nullptr, *recvR = nullptr;
    if (left >= 0) { HIP_OK(hipMalloc(&sendL, msgBytes)); HIP_OK(hipMalloc(&recvL, msgBytes)); HIP_OK(hipMemset(recvL, 0, msgBytes)); }
    if (right >= 0) { HIP_OK(hipMalloc(&sendR, msgBytes)); HIP_OK(hipMalloc(&recvR, msgBytes)); HIP_OK(hipMemset(recvR, 0, msgBytes)); }
    double *redDev = nullptr;
    HIP_OK(hipMalloc((void **)&redDev, 2 * sizeof(double)));

    auto exchange = [&]() {
        NRS_OK_(nrs_slab_pack(ctx, sendL, sendR, msgCap, nullptr)); // (no counts: the call does not wait, the sends follow in stream order)
        // one fixed-size message per neighbour and direction, all four in one group (xGMI is point to point: only the two
        // nearest-neighbour links carry traffic)
        NCCL_OK(ncclGroupStart());
        if (left >= 0) { NCCL_OK(ncclSend(sendL, msgBytes, ncclChar, left, comm, stream)); NCCL_OK(ncclRecv(recvL, msgBytes, ncclChar, left, comm, stream)); }
        if (right >= 0) { NCCL_OK(ncclSend(sendR, msgBytes, ncclChar, right, comm, stream)); NCCL_OK(ncclRecv(recvR, msgBytes, ncclChar, right, comm, stream)); }
        NCCL_OK(ncclGroupEnd());
        NRS_OK_(nrs_slab_unpack(ctx, recvL, recvR, msgCap)); // stream-ordered behind the receives
    };

    uint32_t truncated = 0, lastIters = 0;
    const int maxIters = (halo - 4) / 2; // what the halo width supports (include/nereus_hip.h, nrs_iisph_*)
    for (int s = 0; s < steps; ++s) {
        exchange();
        if (solver != NRS_SOLVER_IISPH) {
            NRS_OK_(nrs_step(ctx, 1));
            continue;
        }
        NRS_OK_(nrs_iisph_predict(ctx));
        int it = 0;
        float rho_avg = 0.f;
        while ((rho_avg - 1000.f) > 1.f || it < 2) { // pressureSolve's loop, on the average over ALL ranks
            if (it >= maxIters) { ++truncated; break; }
            double h[2];
            uint64_t cnt = 0;
            NRS_OK_(nrs_iisph_iterate(ctx, &h[0], &cnt));
            h[1] = (double)cnt;
            HIP_OK(hipMemcpyAsync(redDev, h, sizeof(h), hipMemcpyHostToDevice, stream));
            NCCL_OK(ncclAllReduce(redDev, redDev, 2, ncclDouble, ncclSum, comm, stream));
            HIP_OK(hipMemcpyAsync(h, redDev, sizeof(h), hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            rho_avg = (float)h[0];      // `rho_avg = (SReal)acc; rho_avg /= N` (sph_cuda.cu:818-819)
            rho_avg /= (float)h[1];
            ++it;
        }
        NRS_OK_(nrs_iisph_finish(ctx));
        lastIters = (uint32_t)it;
    }
    exchange(); // drop the halo copies, hand over the last leavers: the first nrs_num_owned() particles are this rank's
    const uint64_t owned = nrs_num_owned(ctx), local = nrs_num_particles(ctx);
    std::vector<float> op(4 * (size_t)local), ov(4 * (size_t)local);
    NRS_OK_(nrs_download(ctx, op.data(), ov.data(), nullptr));
    FILE *o = std::fopen(argv[3], "wb");
    if (!o) die("cannot open the output file");
    const uint32_t oh[4] = {(uint32_t)owned, truncated, lastIters, 0u};
    std::fwrite(oh, 4, 4, o);
    std::fwrite(op.data(), 16, owned, o);
    std::fwrite(ov.data(), 16, owned, o);
    std::fclose(o);
    NRS_OK_(nrs_destroy(ctx));
    NCCL_OK(ncclCommDestroy(comm));
    for (void *p : {sendL, sendR, recvL, recvR, (void *)redDev}) if (p) HIP_OK(hipFree(p));
    HIP_OK(hipStreamDestroy(stream));
    std::printf("slab_rccl rank %d/%d: %d steps, %llu owned, %u truncated\n", rank, world, steps, (unsigned long long)owned, truncated);
    return 0;
}
