// headless.cpp — drives Nereus::SPH / Nereus::IISPH through the class API only (what main.cpp does, minus
// the viewer) and dumps the state for the parity tests.
//
//   headless params  <sesph|iisph> <out.bin>                    constructor-default SphSimParams bytes
//   headless run     <sesph|iisph> <in.bin> <steps> <out.bin>   particles/boundaries from a file
//   headless resume  <sesph|iisph> <in.bin> <steps_a> <steps_b> <ckpt> <out.bin>   run steps_a, saveState, then a NEW
//                                                               solver loadState()s and runs steps_b (boundaries re-set)
//   headless cfl     sesph <in.bin> <steps> <out.bin>           run with setAdaptiveTimestep(true)
//   headless mainscene <sesph|iisph> <steps> <out.bin>          main.cpp:533-553: generateParticleCube + sampleBox +
//                                                               getVbi + updateGpuBoundaries, gravity as given
// in.bin : u32 n, u32 nb, then pos4[n], vel4[n], bi4[nb], vbi[nb] (SReal)
// out.bin: u32 n, u32 nb, u32 iters, u32 sizeof(params), params, pos4[n], vel4[n], pressure[n], bi4[nb], vbi[nb]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "iisph/iisph.h"
#include "pcisph/pcisph.h"
#include "sph.h"
#include <sph_boundary_particles/boundary_forces.h>
#include <sph_boundary_particles/ss.h>

static void die(const char *m) { std::fprintf(stderr, "headless: %s\n", m); std::exit(2); }

static void dump(const char *path, Nereus::SPH *s, unsigned iters, const std::vector<SVec4> &bi, const std::vector<SReal> &vbi)
{
    FILE *f = std::fopen(path, "wb");
    if (!f) die("cannot open output");
    const unsigned n = s->getNumParticles(), nb = (unsigned)bi.size(), ps = sizeof(SphSimParams);
    const SphSimParams P = s->getParams();
    std::fwrite(&n, 4, 1, f); std::fwrite(&nb, 4, 1, f); std::fwrite(&iters, 4, 1, f); std::fwrite(&ps, 4, 1, f);
    std::fwrite(&P, ps, 1, f);
    std::fwrite(s->getHostPos(), sizeof(SReal) * 4, n, f);
    std::fwrite(s->getHostVel(), sizeof(SReal) * 4, n, f);
    std::fwrite(s->getHostPressure(), sizeof(SReal), n, f);
    if (nb) { std::fwrite(bi.data(), sizeof(SVec4), nb, f); std::fwrite(vbi.data(), sizeof(SReal), nb, f); }
    std::fclose(f);
}

int main(int argc, char **argv)
{
    if (argc < 4) die("usage: see source header");
    const std::string mode = argv[1], kind = argv[2];
    if (mode == "vbi") { // headless vbi <bi.f32 (xyzw)> <h> <out.f32>: getVbi (device) then getVbiHost (host cross-check)
        if (argc < 5) die("vbi needs <bi.f32> <h> <out.f32>");
        FILE *f = std::fopen(argv[2], "rb");
        if (!f) die("cannot open input");
        std::fseek(f, 0, SEEK_END);
        const size_t nb = (size_t)std::ftell(f) / sizeof(SVec4);
        std::fseek(f, 0, SEEK_SET);
        std::vector<SVec4> b(nb);
        if (nb && std::fread(b.data(), sizeof(SVec4), nb, f) != nb) die("short input");
        std::fclose(f);
        std::vector<SReal> dev, host;
        const SReal h = (SReal)std::atof(argv[3]);
        sample_spheres::boundary_forces::getVbi(dev, b, h);
        sample_spheres::boundary_forces::getVbiHost(host, b, h);
        FILE *o = std::fopen(argv[4], "wb");
        if (!o) die("cannot open output");
        std::fwrite(dev.data(), sizeof(SReal), nb, o);
        std::fwrite(host.data(), sizeof(SReal), nb, o);
        std::fclose(o);
        return 0;
    }
    const bool iisph = kind == "iisph", pcisph = kind == "pcisph";
    Nereus::SPH *sim = iisph ? (Nereus::SPH *)new Nereus::IISPH() : (pcisph ? (Nereus::SPH *)new Nereus::PCISPH() : new Nereus::SPH());
    sim->_initialize();
    std::vector<SVec4> bi;
    std::vector<SReal> vbi;
    unsigned iters = 0;
    if (mode == "params") {
        dump(argv[3], sim, 0, bi, vbi);
    } else if (mode == "run" || mode == "frames") {
        if (argc < 6) die("run needs <in.bin> <steps> <out.bin>");
        FILE *f = std::fopen(argv[3], "rb");
        if (!f) die("cannot open input");
        unsigned n = 0, nb = 0;
        if (std::fread(&n, 4, 1, f) != 1 || std::fread(&nb, 4, 1, f) != 1) die("short input");
        std::vector<SVec4> pos(n), vel(n);
        bi.resize(nb); vbi.resize(nb);
        if (n && (std::fread(pos.data(), sizeof(SVec4), n, f) != n || std::fread(vel.data(), sizeof(SVec4), n, f) != n)) die("short input");
        if (nb && (std::fread(bi.data(), sizeof(SVec4), nb, f) != nb || std::fread(vbi.data(), sizeof(SReal), nb, f) != nb)) die("short input");
        std::fclose(f);
        for (unsigned i = 0; i < n; ++i) sim->addNewParticle(pos[i], vel[i]);
        if (nb) {
            sim->setNumBoundaries(nb);
            sim->setBi((SReal *)bi.data());
            sim->setVbi(vbi.data());
            sim->updateGpuBoundaries(nb);
        }
        const int steps = std::atoi(argv[4]);
        if (mode == "frames") { // render-loop pattern: update(); draw(latestFrame()) — the frame may lag, never the solver
            sim->setAsyncReadback(true);
            unsigned long long last = 0;
            for (int s = 0; s < steps; ++s) {
                sim->update();
                SUint fn = 0;
                unsigned long long fstep = 0;
                const SReal *frame = sim->latestFrame(&fn, &fstep);
                if (!frame || fn != sim->getNumParticles()) die("latestFrame: no frame");
                if (fstep < last || fstep > (unsigned long long)(s + 1) || fstep + 2 < (unsigned long long)(s + 1)) die("latestFrame: stale or out of order");
                last = fstep;
            }
        } else {
            for (int s = 0; s < steps; ++s) sim->update();
        }
        if (iisph) iters = static_cast<Nereus::IISPH *>(sim)->getLastIterations();
        dump(argv[5], sim, iters, bi, vbi);
    } else if (mode == "resume" || mode == "cfl") {
        FILE *f = std::fopen(argv[3], "rb");
        if (!f) die("cannot open input");
        unsigned n = 0, nb = 0;
        if (std::fread(&n, 4, 1, f) != 1 || std::fread(&nb, 4, 1, f) != 1) die("short input");
        std::vector<SVec4> pos(n), vel(n);
        bi.resize(nb); vbi.resize(nb);
        if (n && (std::fread(pos.data(), sizeof(SVec4), n, f) != n || std::fread(vel.data(), sizeof(SVec4), n, f) != n)) die("short input");
        if (nb && (std::fread(bi.data(), sizeof(SVec4), nb, f) != nb || std::fread(vbi.data(), sizeof(SReal), nb, f) != nb)) die("short input");
        std::fclose(f);
        auto setup = [&](Nereus::SPH *s, bool particles) {
            if (particles) for (unsigned i = 0; i < n; ++i) s->addNewParticle(pos[i], vel[i]);
            if (nb) { s->setNumBoundaries(nb); s->setBi((SReal *)bi.data()); s->setVbi(vbi.data()); s->updateGpuBoundaries(nb); }
        };
        if (mode == "cfl") {
            if (argc < 6) die("cfl needs <in.bin> <steps> <out.bin>");
            setup(sim, true);
            sim->setAdaptiveTimestep(true);
            for (int s = 0; s < std::atoi(argv[4]); ++s) sim->update();
            dump(argv[5], sim, 0, bi, vbi);
        } else {
            if (argc < 8) die("resume needs <in.bin> <steps_a> <steps_b> <ckpt> <out.bin>");
            setup(sim, true);
            for (int s = 0; s < std::atoi(argv[4]); ++s) sim->update();
            if (!sim->saveState(argv[6])) die("saveState failed");
            Nereus::SPH *again = iisph ? (Nereus::SPH *)new Nereus::IISPH() : new Nereus::SPH();
            again->_initialize();
            if (!again->loadState(argv[6])) die("loadState failed");
            setup(again, false);
            for (int s = 0; s < std::atoi(argv[5]); ++s) again->update();
            if (iisph) iters = static_cast<Nereus::IISPH *>(again)->getLastIterations();
            dump(argv[7], again, iters, bi, vbi);
            delete again;
        }
    } else if (mode == "mainscene") {
        if (argc < 5) die("mainscene needs <steps> <out.bin>");
        sim->generateParticleCube(make_SVec4(-0.4f, 0.04f, 0.5f, 1.f), make_SVec4(0.5f, 0.5f, 0.5f, 1.f), make_SVec4(0, 0, 0, 0));
        if (std::getenv("NEREUS_MAIN_GRAVITY_OFF")) sim->setGravity(0.0); // main.cpp:538
        sample_spheres::ss::sampleBox(bi, make_SVec3(-1, -1, -1), make_SVec3(3.f, 3.f, 3.f), 0.02);
        sample_spheres::boundary_forces::getVbi(vbi, bi, sim->getInteractionRadius());
        sim->setNumBoundaries(bi.size());
        sim->setBi((SReal *)bi.data());
        sim->setVbi(vbi.data());
        sim->updateGpuBoundaries(bi.size());
        const int steps = std::atoi(argv[3]);
        for (int s = 0; s < steps; ++s) sim->update();
        if (iisph) iters = static_cast<Nereus::IISPH *>(sim)->getLastIterations();
        dump(argv[4], sim, iters, bi, vbi);
    } else {
        die("unknown mode");
    }
    sim->_finalize();
    delete sim;
    return 0;
}
