/*
 * sph_run -- the reference's time-stepping driver on top of the C ABI (include/sph_amd.h).
 *
 * Keeps what main() of the reference does around the hot path (WCXSPH3D.cpp:439-499,
 * WCXSPH.cpp:411-471, WCSPH.cpp:412-474): InitSPH, the t=0 Forces call, the settings summary,
 * the Tecplot .plt strand file, the per-frame convergence line and the wall-clock print.
 * The reference takes a settings file name on its command line and ignores it
 * (makefile:15-19, Sample_Input.txt); here it is read.
 *
 * Settings grammar (Sample_Input.txt:1-12): one "key value" per line, '#' starts a comment.
 *   default_params? yes|no   yes: ignore every other physics key, run the program as shipped
 *   outfile NAME             NAME.plt and NAME_Settings.txt          (reference: Test)
 *   nframes N                Nframe                                  (reference: 2000)
 *   npframe N                Ndtframe                                (reference: 10)
 *   dt X                     initial time step                       (reference: 0.0002)
 *   psize X                  Pstep                                   (reference: 0.1)
 *   rho_ref X                rho0                                    (reference: 1000)
 *   viscosity X              artificial-viscosity alpha              (reference: 0.025)
 *   gravity X                gravitational strength                  (reference: 9.81)
 *   Bulk_mod X               no counterpart: the reference derives the Tait modulus from the
 *                            speed of sound (B = rho0*Cs^2/gam, WCXSPH3D.cpp:54-56); accepted
 *                            and ignored with a notice
 *   simtype Box|Droplet|Jet  Box is the only initialiser the reference has; Droplet / Jet add a
 *                            sphere / vertical cylinder of fluid on the block's lattice spacing:
 *                            drop_radius R, drop_centre X Y [Z], drop_height L (Jet),
 *                            drop_velocity VX VY [VZ]
 * Extensions of this driver:
 *   program WCSPH|WCXSPH|WCXSPH3D   which reference program to run  (default WCXSPH)
 *   npart NX NY [NZ], box BX BY [BZ], start SX SY [SZ], cs X
 *   reinit 0|1               per-frame DensityReinit                (default 1)
 *   device N                 HIP device ordinal
 *   frame_stride K           write every K-th frame to the .plt     (default 1: all, as the reference)
 *   threads N                host threads that format a frame       (default: the machine's, at most 16)
 *   transport rccl|host      several GPUs: how halo bytes travel     (default rccl; host = staged through
 *                            this program's sockets, for ranks that share a GPU)
 *   migrate 0|1              several GPUs: particles change owner as they cross a slab plane (default 1)
 *
 * Several GPUs: start one process per GPU with the rank, the number of ranks and the local rank
 * in the environment (RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run sets them, or
 * OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK, or SLURM_PROCID / SLURM_NTASKS / SLURM_LOCALID) and
 * the address of rank 0 in MASTER_ADDR, its port in SPH_WIRE_PORT (default MASTER_PORT + 1, or
 * 29617).  The case is cut into x-slabs of equal weighted particle count; every rank sets up and
 * steps its own slab; rank 0 gathers the rows of each written frame and writes the one .plt file.
 * The rows of a frame are formatted by several threads into memory and written in order, and the
 * file is written while the GPU computes the next frame; the bytes are the same as from the
 * reference's single `f1 << ... << std::endl` loop.
 */
#include "../../include/sph_amd.h"
#include "wire.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <cstdio>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Settings {
    bool defaults = true;
    std::string outfile = "Test";
    int nframes = 2000, npframe = 10;
    int program = SPH_PROGRAM_WCXSPH;
    int reinit = 1, device = -1;
    int frame_stride = 1, threads = 0;
    int migrate = 1;
    std::string transport = "rccl";
    bool have_npart = false, have_box = false, have_start = false;
    double dt = 0, psize = 0, rho_ref = 0, viscosity = 0, gravity = 0, cs = 0;
    double npart[3] = {0, 0, 0}, box[3] = {0, 0, 0}, start[3] = {0, 0, 0};
    int simtype = SPH_SIMTYPE_BOX;
    double drop_radius = 0, drop_height = 0, drop_centre[3] = {0, 0, 0}, drop_v[3] = {0, 0, 0};
};

bool read_settings(const char* path, Settings& s) {
    std::ifstream in(path);
    if (!in.is_open()) return false;
    std::string line;
    while (std::getline(in, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream ls(line);
        std::string key;
        if (!(ls >> key)) continue;
        if (key == "default_params?") { std::string v; ls >> v; s.defaults = (v == "yes" || v == "Yes" || v == "1"); }
        else if (key == "outfile") ls >> s.outfile;
        else if (key == "nframes") ls >> s.nframes;
        else if (key == "npframe") ls >> s.npframe;
        else if (key == "dt") ls >> s.dt;
        else if (key == "psize") ls >> s.psize;
        else if (key == "rho_ref") ls >> s.rho_ref;
        else if (key == "viscosity") ls >> s.viscosity;
        else if (key == "gravity") ls >> s.gravity;
        else if (key == "cs") ls >> s.cs;
        else if (key == "Bulk_mod") std::cout << "Note: Bulk_mod has no counterpart (B follows from the speed of sound); ignored." << std::endl;
        else if (key == "simtype") {
            std::string v; ls >> v;
            if (v == "Box") s.simtype = SPH_SIMTYPE_BOX;
            else if (v == "Droplet") s.simtype = SPH_SIMTYPE_DROPLET;
            else if (v == "Jet") s.simtype = SPH_SIMTYPE_JET;
            else { std::cerr << "Unknown simtype " << v << std::endl; return false; }
        }
        else if (key == "drop_radius") ls >> s.drop_radius;
        else if (key == "drop_height") ls >> s.drop_height;
        else if (key == "drop_centre") { for (int d = 0; d < 3 && (ls >> s.drop_centre[d]); ++d) {} }
        else if (key == "drop_velocity") { for (int d = 0; d < 3 && (ls >> s.drop_v[d]); ++d) {} }
        else if (key == "program") {
            std::string v; ls >> v;
            if (v == "WCSPH") s.program = SPH_PROGRAM_WCSPH;
            else if (v == "WCXSPH") s.program = SPH_PROGRAM_WCXSPH;
            else if (v == "WCXSPH3D") s.program = SPH_PROGRAM_WCXSPH3D;
            else { std::cerr << "Unknown program " << v << std::endl; return false; }
        }
        else if (key == "npart") { s.have_npart = true; for (int d = 0; d < 3 && (ls >> s.npart[d]); ++d) {} }
        else if (key == "box") { s.have_box = true; for (int d = 0; d < 3 && (ls >> s.box[d]); ++d) {} }
        else if (key == "start") { s.have_start = true; for (int d = 0; d < 3 && (ls >> s.start[d]); ++d) {} }
        else if (key == "reinit") ls >> s.reinit;
        else if (key == "device") ls >> s.device;
        else if (key == "frame_stride") { ls >> s.frame_stride; if (s.frame_stride < 1) s.frame_stride = 1; }
        else if (key == "threads") ls >> s.threads;
        else if (key == "migrate") ls >> s.migrate;
        else if (key == "transport") {
            ls >> s.transport;
            if (s.transport != "rccl" && s.transport != "host") { std::cerr << "Unknown transport " << s.transport << std::endl; return false; }
        }
        else std::cout << "Note: unknown settings key '" << key << "' ignored." << std::endl;
    }
    return true;
}

struct HostState {
    int dim = 0;
    size_t n = 0;
    std::vector<double> x[3], v[3], V[3], f[3];
    std::vector<float> rho, p, Rrho, m;
    sph_fields fields() {
        sph_fields h = {};
        for (int d = 0; d < dim; ++d) { h.x[d] = x[d].data(); h.v[d] = v[d].data(); h.V[d] = V[d].data(); h.f[d] = f[d].data(); }
        h.rho = rho.data(); h.p = p.data(); h.Rrho = Rrho.data(); h.m = m.data();
        return h;
    }
    void resize(int dim_, size_t n_) {
        dim = dim_; n = n_;
        for (int d = 0; d < dim; ++d) { x[d].assign(n, 0); v[d].assign(n, 0); V[d].assign(n, 0); f[d].assign(n, 0); }
        rho.assign(n, 0); p.assign(n, 0); Rrho.assign(n, 0); m.assign(n, 0);
    }
    double norm(const std::vector<double>* a, size_t i) const {
        double s = a[0][i] * a[0][i] + a[1][i] * a[1][i];
        if (dim == 3) s = s + a[2][i] * a[2][i];
        return sqrt(s);
    }
};

/* write_settings (WCXSPH3D.cpp:381-406): same lines, same default stream precision */
void write_settings(const std::string& name, int nframes, int npframe, double dt, double pstep, double mass, double rho0,
                    double H, double gravity, uint64_t bound_parts, uint64_t simpts) {
    std::ofstream fp(name.c_str(), std::ios::out);
    if (fp.is_open()) {
        fp << "SIMULATION PARAMETERS:" << std::endl;
        fp << "\tNumber of frames (" << nframes << ")" << std::endl;
        fp << "\tSteps per frame (" << npframe << ")" << std::endl;
        fp << "\tTime step (" << dt << ")" << std::endl;
        fp << "\tParticle Spacing (" << pstep << ")" << std::endl;
        fp << "\tParticle Mass (" << mass << ")" << std::endl;
        fp << "\tReference density (" << rho0 << ")" << std::endl;
        fp << "\tSupport Radius (" << H << ")" << std::endl;
        fp << "\tGravitational strength (" << gravity << ")" << std::endl;
        fp << "\tNumber of boundary points (" << bound_parts << ")" << std::endl;
        fp << "\tNumber of simulation points (" << simpts << ")" << std::endl;
        fp << "\tIntegrator type (Newmark_Beta)" << std::endl;
        fp.close();
    } else {
        std::cout << "Error opening the output file." << std::endl;
        exit(-1);
    }
}

/* write_frame_data (WCXSPH3D.cpp:408-437; WCSPH.cpp:378-410 adds DATAPACKING=POINT and prints
 * the components in its NaN report).  One row: x y [z] |v| |f| rho p, fixed, three decimals --
 * printf's "%.3f" produces the characters `std::fixed << std::setprecision(3)` does. */
void format_rows(const HostState& s, size_t lo, size_t hi, std::string& out) {
    char buf[256];
    out.clear();
    out.reserve((hi - lo) * 56);
    for (size_t i = lo; i < hi; ++i) {
        int n = 0;
        for (int d = 0; d < s.dim; ++d) n += snprintf(buf + n, sizeof buf - n, "%.3f ", s.x[d][i]);
        n += snprintf(buf + n, sizeof buf - n, "%.3f %.3f %.3f %.3f\n", s.norm(s.v, i), s.norm(s.f, i), (double)s.rho[i],
                      (double)s.p[i]);
        out.append(buf, (size_t)n);
    }
}

/* first fluid particle with a NaN in x, v or f, or n if there is none */
size_t first_broken(const HostState& s, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i)
        for (int d = 0; d < s.dim; ++d)
            if (s.x[d][i] != s.x[d][i] || s.v[d][i] != s.v[d][i] || s.f[d][i] != s.f[d][i]) return i;
    return s.n;
}

/* Returns false after the reference's NaN report (the caller stops the run from its main thread). */
bool write_frame(const HostState& s, std::ofstream& fp, float t, uint64_t bound_parts, int program, int nthreads) {
    /* the reference stops at the first broken fluid row, after writing the rows before it */
    const size_t bad = first_broken(s, (size_t)bound_parts, s.n);
    const size_t stop = bad < s.n ? bad : s.n;
    /* chunks [0, bound) and [bound, stop), each cut into pieces formatted in parallel */
    std::vector<std::pair<size_t, size_t> > pieces;
    const size_t per = std::max<size_t>(65536, (stop + nthreads - 1) / std::max(nthreads, 1));
    size_t wall_pieces = 0;
    for (size_t lo = 0; lo < bound_parts; lo += per) { pieces.push_back(std::make_pair(lo, std::min<size_t>(lo + per, bound_parts))); ++wall_pieces; }
    for (size_t lo = bound_parts; lo < stop; lo += per) pieces.push_back(std::make_pair(lo, std::min(lo + per, stop)));
    std::vector<std::string> text(pieces.size());
    std::vector<std::thread> pool;
    const int workers = std::max(1, std::min<int>(nthreads, (int)pieces.size()));
    for (int w = 0; w < workers; ++w)
        pool.emplace_back([&, w]() {
            for (size_t k = (size_t)w; k < pieces.size(); k += (size_t)workers) format_rows(s, pieces[k].first, pieces[k].second, text[k]);
        });
    for (auto& th : pool) th.join();
    fp << "ZONE t=\"Boundary Data\", STRANDID=1, SOLUTIONTIME=" << t << std::endl;
    for (size_t k = 0; k < wall_pieces; ++k) fp.write(text[k].data(), (std::streamsize)text[k].size());
    fp << "ZONE t=\"Particle Data\", STRANDID=2, SOLUTIONTIME=" << t;
    if (program == SPH_PROGRAM_WCSPH) fp << ", DATAPACKING=POINT";
    fp << std::endl;
    for (size_t k = wall_pieces; k < pieces.size(); ++k) fp.write(text[k].data(), (std::streamsize)text[k].size());
    if (bad < s.n) {
        const size_t i = bad;
        std::cerr << std::endl << "Simulation is broken. A value is nan." << std::endl;
        std::cerr << "Broken line..." << std::endl;
        std::cerr << std::fixed << std::setprecision(3);
        if (program == SPH_PROGRAM_WCSPH) {
            for (int d = 0; d < s.dim; ++d) std::cerr << s.x[d][i] << " ";
            for (int d = 0; d < s.dim; ++d) std::cerr << s.v[d][i] << " ";
            for (int d = 0; d < s.dim; ++d) std::cerr << s.f[d][i] << " ";
            std::cerr << s.rho[i] << " " << s.p[i] << std::endl;
        } else {
            for (int d = 0; d < s.dim; ++d) std::cerr << s.x[d][i] << " ";
            std::cerr << s.norm(s.v, i) << " " << s.norm(s.f, i) << " " << s.rho[i] << " " << s.p[i] << std::endl;
        }
        fp.close();
        return false;
    }
    return true;
}

/* ---- several GPUs ------------------------------------------------------------------------- */

int env_int(const char* const* names, int fallback) {
    for (; *names; ++names)
        if (const char* v = getenv(*names)) return atoi(v);
    return fallback;
}

/* x-planes p[0] = -inf < ... < p[nranks] = +inf that give the slabs about equal work; a wall
 * particle counts `wall_weight` of a fluid one (pairs of two resting wall particles are never
 * evaluated) */
bool choose_planes(const sph_case& cs, int nranks, std::vector<double>& planes, double wall_weight = 0.4, int nbins = 16384) {
    std::vector<uint64_t> fl((size_t)nbins), wl((size_t)nbins);
    if (sph_case_histogram_x(&cs, nbins, fl.data(), wl.data()) != SPH_OK) return false;
    std::vector<double> cum((size_t)nbins);
    double run = 0;
    for (int b = 0; b < nbins; ++b) {
        run += (double)fl[(size_t)b] + wall_weight * (double)wl[(size_t)b];
        cum[(size_t)b] = run;
    }
    planes.assign(1, -1e300);
    for (int r = 1; r < nranks; ++r) {
        const double target = run * r / nranks;
        int b = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        b = std::min(b + 1, nbins);
        planes.push_back(cs.box[0] * (double)b / (double)nbins);
    }
    planes.push_back(1e300);
    for (size_t k = 0; k + 1 < planes.size(); ++k)
        if (!(planes[k + 1] > planes[k])) return false;
    return true;
}

/* the host-staged transport of include/sph_amd.h over this program's sockets */
int wire_sendrecv(void* user, const void* sL, uint64_t nsL, void* rL, uint64_t nrL, const void* sR, uint64_t nsR, void* rR,
                  uint64_t nrR, void*) {
    return ((Wire*)user)->sendrecv(sL, nsL, rL, nrL, sR, nsR, rR, nrR) ? 0 : 1;
}
int wire_allreduce(void* user, double* vals, int32_t n, int32_t op, void*) { return ((Wire*)user)->allreduce(vals, n, op) ? 0 : 1; }

/* one rank's rows of a frame as they travel to rank 0: count, global indices, then the fields */
void pack_rows(const HostState& s, const std::vector<uint32_t>& gids, std::vector<char>& out) {
    const uint64_t n = s.n;
    out.resize(8 + n * 4 + n * 8 * (size_t)(3 * s.dim) + n * 4 * 2);
    char* p = out.data();
    auto put = [&p](const void* src, size_t bytes) { memcpy(p, src, bytes); p += bytes; };
    put(&n, 8);
    put(gids.data(), n * 4);
    for (int d = 0; d < s.dim; ++d) { put(s.x[d].data(), n * 8); put(s.v[d].data(), n * 8); put(s.f[d].data(), n * 8); }
    put(s.rho.data(), n * 4);
    put(s.p.data(), n * 4);
}
/* false when the message is not what pack_rows makes for a whole of `whole.n` rows: its length
 * does not follow from its count, or an index names no row */
bool unpack_rows(const char* p, size_t bytes, int dim, HostState& whole) {
    if (bytes < 8) return false;
    uint64_t n;
    memcpy(&n, p, 8);
    if (n > whole.n || bytes != 8 + n * 4 + n * 8 * (uint64_t)(3 * dim) + n * 4 * 2) return false;
    p += 8;
    const uint32_t* g = (const uint32_t*)p;
    for (uint64_t i = 0; i < n; ++i)
        if (g[i] >= whole.n) return false;
    p += n * 4;
    for (int d = 0; d < dim; ++d) {
        std::vector<double>* dst[3] = {&whole.x[d], &whole.v[d], &whole.f[d]};
        for (int q = 0; q < 3; ++q) {
            const double* src = (const double*)p;
            for (uint64_t i = 0; i < n; ++i) (*dst[q])[g[i]] = src[i];
            p += n * 8;
        }
    }
    const float* r = (const float*)p;
    for (uint64_t i = 0; i < n; ++i) whole.rho[g[i]] = r[i];
    p += n * 4;
    const float* pr = (const float*)p;
    for (uint64_t i = 0; i < n; ++i) whole.p[g[i]] = pr[i];
    return true;
}

/* the sockets of this process, for die(): a rank that gives up closes them, so that the others
 * fail in their next exchange instead of waiting for it */
Wire* g_wire = nullptr;

void die(sph_ctx* ctx, const char* what) {
    std::cerr << what << ": " << sph_last_error(ctx) << std::endl;
    if (g_wire) g_wire->shutdown_all();
    exit(-1);
}

}  // namespace

int main(int argc, char* argv[]) {
    auto t1 = std::chrono::high_resolution_clock::now();
    Settings st;
    if (argc > 1 && !read_settings(argv[1], st)) {
        std::cerr << "Error opening the settings file." << std::endl;
        exit(-1);
    }

    sph_case cs;
    if (sph_case_default(st.program, &cs) != SPH_OK) exit(-1);
    if (!st.defaults) {
        if (st.dt > 0) cs.dt0 = st.dt;
        if (st.psize > 0) cs.pstep = st.psize;
        cs.rho0 = st.rho_ref;
        cs.alpha = st.viscosity;
        cs.gravity = st.gravity;
        cs.Cs = st.cs;
        for (int d = 0; d < 3; ++d) {
            if (st.have_npart && st.npart[d] > 0) cs.npart[d] = (int)st.npart[d];
            if (st.have_box && st.box[d] > 0) cs.box[d] = st.box[d];
            if (st.have_start) cs.start[d] = st.start[d];
        }
        cs.simtype = st.simtype;
        cs.drop_radius = st.drop_radius;
        cs.drop_height = st.drop_height;
        for (int d = 0; d < 3; ++d) { cs.drop_centre[d] = st.drop_centre[d]; cs.drop_v[d] = st.drop_v[d]; }
    }
    sph_params prm;
    double simmass = 0, boundmass = 0;
    if (sph_case_params(&cs, &prm, &simmass, &boundmass) != SPH_OK) {
        std::cerr << "Invalid simulation definition." << std::endl;
        exit(-1);
    }
    const uint64_t npts_all = prm.npts, bound_parts = prm.bound_parts;
    const uint64_t simpts = npts_all - bound_parts;

    /* one process per GPU: who am I (read before anything touches a GPU) */
    static const char* const k_rank[] = {"RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_PROCID", nullptr};
    static const char* const k_size[] = {"WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS", nullptr};
    static const char* const k_local[] = {"LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", nullptr};
    const int rank = env_int(k_rank, 0), nranks = std::max(1, env_int(k_size, 1)), local_rank = env_int(k_local, 0);
    const bool root = rank == 0, multi = nranks > 1;
    Wire wire;
    g_wire = &wire;
    if (multi) {
        static const char* const k_port[] = {"SPH_WIRE_PORT", nullptr};
        static const char* const k_mport[] = {"MASTER_PORT", nullptr};
        int port = env_int(k_port, 0);
        if (port <= 0) port = env_int(k_mport, 29616) + 1;
        const char* addr = getenv("MASTER_ADDR");
        if (!wire.open(rank, nranks, addr ? addr : "127.0.0.1", port)) {
            std::cerr << "rank " << rank << ": " << wire.error << std::endl;
            exit(-1);
        }
        /* nothing a rank sends is larger than all the particles' full records (240 bytes each
         * when they migrate, 84 in a frame's rows) */
        wire.max_message = 4096 + npts_all * 256;
    }
    prm.device = st.device >= 0 ? st.device : (multi ? local_rank : -1);

    /* InitSPH (WCXSPH3D.cpp:306-379): the whole case on one GPU, this rank's slab otherwise */
    if (root) std::cout << "Initialising simulation with " << simpts << " particles" << std::endl;
    HostState hs, hs_alt; /* two sets: one is written to the file while the other is filled */
    HostState local;      /* several GPUs: this rank's rows */
    std::vector<uint32_t> gids;
    int nthreads = st.threads > 0 ? st.threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (root) {
        hs.resize(prm.dim, (size_t)npts_all);
        hs_alt.resize(prm.dim, (size_t)npts_all);
    }
    if (multi) {
        std::vector<double> planes;
        if (!choose_planes(cs, nranks, planes)) {
            std::cerr << "Cannot cut " << nranks << " slabs out of this case." << std::endl;
            exit(-1);
        }
        uint64_t n_own = 0;
        if (sph_case_init_range(&cs, planes[(size_t)rank], planes[(size_t)rank + 1], nullptr, nullptr, &n_own) != SPH_OK || n_own == 0) {
            std::cerr << "rank " << rank << " owns no particle" << std::endl;
            exit(-1);
        }
        local.resize(prm.dim, (size_t)n_own);
        gids.resize((size_t)n_own);
        sph_fields hl = local.fields();
        if (sph_case_init_range(&cs, planes[(size_t)rank], planes[(size_t)rank + 1], &hl, gids.data(), &n_own) != SPH_OK) exit(-1);
        const uint64_t room = std::max<uint64_t>(1u << 16, n_own / 8);
        prm.npts_global = npts_all;
        prm.npts = n_own;
        prm.rank = rank;
        prm.nranks = nranks;
        prm.capacity = n_own + 4 * room; /* growth by migration 2*room, each side's halo room */
        prm.migrate = st.migrate ? 1 : 0;
        prm.slab_lo = planes[(size_t)rank];
        prm.slab_hi = planes[(size_t)rank + 1];
    } else {
        sph_fields h = hs.fields();
        if (sph_case_init(&cs, &h) != SPH_OK) exit(-1);
    }
    if (root) std::cout << "Total Particles: " << npts_all << std::endl;

    sph_ctx* ctx = nullptr;
    if (sph_create(&prm, &ctx) != SPH_OK) die(ctx, "sph_create");
    {
        sph_fields h = multi ? local.fields() : hs.fields();
        if (sph_upload(ctx, SPH_STATE_N, &h) != SPH_OK) die(ctx, "sph_upload");
        if (sph_upload(ctx, SPH_STATE_NP1, &h) != SPH_OK) die(ctx, "sph_upload"); /* the particlesh copy, :450-451 */
    }
    sph_transport wire_tr;
    if (multi) {
        if (sph_upload_global_ids(ctx, gids.data()) != SPH_OK) die(ctx, "sph_upload_global_ids");
        if (st.transport == "rccl") {
            char id[128];
            memset(id, 0, sizeof id);
            if (root && sph_rccl_unique_id(id) != SPH_OK) {
                std::cerr << "sph_rccl_unique_id failed (is librccl.so.1 on the loader path?)" << std::endl;
                exit(-1);
            }
            if (!wire.bcast(id, sizeof id)) { std::cerr << wire.error << std::endl; exit(-1); }
            if (sph_use_rccl(ctx, id) != SPH_OK) die(ctx, "sph_use_rccl");
        } else {
            memset(&wire_tr, 0, sizeof wire_tr);
            wire_tr.struct_size = sizeof wire_tr;
            wire_tr.user = &wire;
            wire_tr.host_buffers = 1;
            wire_tr.sendrecv = wire_sendrecv;
            wire_tr.allreduce = wire_allreduce;
            if (sph_set_transport(ctx, &wire_tr) != SPH_OK) die(ctx, "sph_set_transport");
        }
    }

    if (sph_forces(ctx, SPH_STATE_N) != SPH_OK) die(ctx, "sph_forces");       /* :456 */
    if (sph_zero_boundary_forces(ctx, SPH_STATE_N) != SPH_OK) die(ctx, "sph_zero_boundary_forces"); /* :457-458 */

    if (root)
        write_settings(st.outfile + "_Settings.txt", st.nframes, st.npframe, cs.dt0, cs.pstep, simmass, prm.rho0, prm.H,
                       -prm.g, bound_parts, simpts);

    /* One frame's rows into `dst` (rank 0; all particles, original order): a download on one
     * GPU; otherwise every rank downloads what it owns now and rank 0 places the rows by
     * their global indices. */
    auto fetch_frame = [&](int which, HostState* dst) {
        if (!multi) {
            sph_fields hf = dst->fields();
            if (sph_download(ctx, which, &hf) != SPH_OK) die(ctx, "sph_download");
            return;
        }
        uint64_t n_now = 0;
        if (sph_local_count(ctx, &n_now, nullptr, nullptr) != SPH_OK) die(ctx, "sph_local_count");
        if (n_now != local.n) {
            local.resize(prm.dim, (size_t)n_now);
            gids.resize((size_t)n_now);
        }
        sph_fields hl = local.fields();
        if (sph_download(ctx, which, &hl) != SPH_OK) die(ctx, "sph_download");
        if (sph_download_global_ids(ctx, gids.data()) != SPH_OK) die(ctx, "sph_download_global_ids");
        std::vector<char> mine;
        pack_rows(local, gids, mine);
        std::vector<std::vector<char> > all;
        if (!wire.gather(mine.data(), mine.size(), &all)) { std::cerr << wire.error << std::endl; exit(-1); }
        if (root) {
            bool good = unpack_rows(mine.data(), mine.size(), prm.dim, *dst);
            for (int k = 1; k < nranks && good; ++k) good = unpack_rows(all[(size_t)k].data(), all[(size_t)k].size(), prm.dim, *dst);
            if (!good) {
                std::cerr << "a rank's frame rows do not fit the run (length or index out of range)" << std::endl;
                wire.shutdown_all();
                exit(-1);
            }
        }
    };

    std::ofstream f1;
    if (root) f1.open((st.outfile + ".plt").c_str(), std::ios::out);
    bool opened = !root || f1.is_open();
    if (multi) { /* everybody learns whether rank 0 has its file */
        double v[1] = {opened ? 0.0 : 1.0};
        if (!wire.allreduce(v, 1, 0)) { std::cerr << wire.error << std::endl; exit(-1); }
        opened = v[0] == 0.0;
    }
    if (opened) {
        if (root) {
            f1 << std::fixed << std::setprecision(3);
            std::cout << "Starting simulation..." << std::endl;
            f1 << (prm.dim == 3 ? "VARIABLES = x, y, z, V, F, rho, P" : "VARIABLES = x, y, V, F, rho, P") << std::endl;
        }
        std::atomic<bool> broken(false); /* set by the writer after the reference's NaN report */
        fetch_frame(SPH_STATE_N, &hs);
        if (root && !write_frame(hs, f1, 0.f, bound_parts, st.program, nthreads)) broken = true;

        sph_scalars sc;
        if (sph_get_scalars(ctx, &sc) != SPH_OK) die(ctx, "sph_get_scalars");
        HostState* fill = &hs;      /* the set the next download goes to */
        HostState* busy = &hs_alt;  /* the set the writer thread is reading, if one is running */
        std::thread writer;
        /* The reference stops inside write_frame_data at the first NaN row (WCXSPH3D.cpp:422-431).
         * Here the file is written by a second thread while the next frame is computed: that thread
         * only raises a flag; the main thread sees it once the writer has been joined, tells the
         * other ranks, releases the context and exits -1. */
        auto stop_if_broken = [&]() {
            int stop = broken ? 1 : 0;
            if (multi) {
                double v[1] = {(double)stop};
                if (!wire.allreduce(v, 1, 0)) { std::cerr << wire.error << std::endl; exit(-1); }
                stop = v[0] != 0.0;
            }
            if (stop) {
                sph_destroy(ctx);
                exit(-1);
            }
        };
        stop_if_broken();
        for (int frame = 1; frame <= st.nframes; ++frame) {
            if (root)
                std::cout << "Frame Number: " << frame << "\tError: " << log10(sqrt(sc.errsum / (1.0 * npts_all))) - sc.logbase
                          << std::endl;
            if (sph_step(ctx, st.npframe, &sc) != SPH_OK) die(ctx, "sph_step");
            if (st.reinit && sph_density_reinit(ctx) != SPH_OK) die(ctx, "sph_density_reinit");
            if (frame % st.frame_stride != 0 && frame != st.nframes) continue;
            fetch_frame(SPH_STATE_NP1, fill);
            if (writer.joinable()) writer.join(); /* frames go to the file in order */
            stop_if_broken();
            if (root) {
                const float tframe = sc.t;
                HostState* job = fill;
                writer = std::thread([&f1, &broken, job, tframe, bound_parts, &st, nthreads]() {
                    if (!write_frame(*job, f1, tframe, bound_parts, st.program, nthreads)) broken = true;
                });
            }
            std::swap(fill, busy);
        }
        if (writer.joinable()) writer.join();
        stop_if_broken();
        if (root) f1.close();
    } else {
        if (root) std::cerr << "Error opening frame output file." << std::endl;
        exit(-1);
    }
    sph_destroy(ctx);

    if (root) {
        std::cout << "Simulation complete. Output is available in /Output!" << std::endl;
        auto t2 = std::chrono::high_resolution_clock::now();
        auto duration = std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count();
        std::cout << "Time taken:\t" << duration / 1e06 << " seconds" << std::endl;
    }
    return 0;
}
