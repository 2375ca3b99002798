// cli_common.h -- flag parsing and small helpers shared by gnnpe_main (GNN-PE/src/main.cpp mirror) and
// gnnpge_main (GNN-PGE/src/main.cpp mirror).  Both reference programs take the same eight CLI11 flags.
#pragma once

#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gnnpe_hip.h"
#include "graph_loader.h"

using Clock = std::chrono::steady_clock;

namespace cli {

struct Options {
    const char *tool = "gnnpe_main";
    // the reference's flags and defaults (main.cpp:40-56, custom.h:45-50)
    std::string dataset_path = "../Test/";
    std::string data_graph = "../Test/data_graph.graph";
    std::string query_graph = "../Test/query_graph.graph";
    std::string mode = "offline";
    std::string answers = "MAX";
    uint32_t partition_num = 5, path_length = 2, vde_dim = 2;
    // extensions
    int gpus = 1;
    uint64_t chunk_paths = 8ull << 20;  // 8M paths per pass: small enough to pipeline render, copy-back and file writes
    bool allow_large = false, timing = false, sidecars = false, write_index = false;
    bool strict = false;  // refuse a graph file with a duplicate `e` line (the reference loads it as it is: graph.cpp:211-218)
    bool same_device = false;  // testing aid: all --gpus contexts on device 0 (halo by device copies: RCCL needs distinct GPUs)
    std::string transport = "rccl";  // --gpus N > 1: "rccl" (ncclSend/ncclRecv over xGMI) or "copy" (device-to-device copies)
    // --transport given explicitly: take the slab path (host/slab_offline.cpp) even with --gpus 1 -- a 1-rank communicator
    // whose exchanges go through librccl (self ncclSend/ncclRecv), i.e. the N > 1 code on a single-GPU box
    bool transport_explicit = false;
};

static const char *g_tool = "gnnpe_main";

inline double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

// seconds since the kernel started this process (exec, dynamic loading, the libraries' static constructors -- what a caller's
// wall-clock holds in front of main()); 10 ms resolution (/proc/self/stat field 22 against CLOCK_BOOTTIME); < 0 when unknown
inline double secs_since_process_start()
{
    FILE *f = fopen("/proc/self/stat", "r");
    if (!f) return -1.0;
    char buf[2048];
    const size_t nr = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[nr] = 0;
    const char *p = strrchr(buf, ')');  // (the command name may hold blanks)
    if (!p) return -1.0;
    unsigned long long start = 0;
    int field = 2;
    for (p++; *p && field < 22;) {
        while (*p == ' ') p++;
        field++;
        if (field == 22) {
            start = strtoull(p, nullptr, 10);
            break;
        }
        while (*p && *p != ' ') p++;
    }
    struct timespec ts;
    if (!start || clock_gettime(CLOCK_BOOTTIME, &ts) != 0) return -1.0;
    return (double)ts.tv_sec + ts.tv_nsec * 1e-9 - (double)start / (double)sysconf(_SC_CLK_TCK);
}

[[noreturn]] inline void die(const std::string &msg, int code = 1)
{
    fprintf(stderr, "%s: %s\n", g_tool, msg.c_str());
    exit(code);
}

inline void check(int rc, const char *what)
{
    if (rc != 0) die(std::string(what) + ": " + gnnpe_last_error());
}

inline bool parse_u32(const std::string &s, uint32_t *v)
{
    if (s.empty()) return false;
    char *end = nullptr;
    unsigned long long x = strtoull(s.c_str(), &end, 10);
    if (*end || x > 0xFFFFFFFFull) return false;
    *v = (uint32_t)x;
    return true;
}

// CLI11-style parsing of `-x v`, `-xv`, `--long v`, `--long=v`
inline Options parse_args(int argc, char **argv, const char *tool = "gnnpe_main")
{
    Options o;
    o.tool = tool;
    g_tool = tool;
    struct Opt { char s; const char *l; } opts[] = {{'f', "file"}, {'d', "data"}, {'q', "query"}, {'m', "mode"},
                                                     {'p', "partition"}, {'l', "length"}, {'e', "embedding"}, {'n', "answers"}};
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i], val;
        char key = 0;
        if (a == "-h" || a == "--help") {
            printf("%s -f <dataset dir/> -d <data.graph> -m offline -p <partitions> [-l 2] [-e 2]\n"
                   "           [--gpus N] [--transport rccl|copy] [--chunk PATHS] [--index] [--sidecars] [--timing] [--allow-large]\n", tool);
            exit(0);
        }
        if (a == "--gpus" || a == "--chunk") {
            if (i + 1 >= argc) die(a + " needs a value");
            uint64_t v = strtoull(argv[++i], nullptr, 10);
            if (a == "--gpus") o.gpus = (int)v; else o.chunk_paths = v;
            continue;
        }
        if (a == "--transport") {
            if (i + 1 >= argc) die(a + " needs a value");
            o.transport = argv[++i];
            if (o.transport != "rccl" && o.transport != "copy") die("--transport must be rccl or copy");
            o.transport_explicit = true;
            continue;
        }
        if (a == "--allow-large") { o.allow_large = true; continue; }
        if (a == "--timing") { o.timing = true; continue; }
        if (a == "--sidecars") { o.sidecars = true; continue; }
        if (a == "--index") { o.write_index = true; continue; }
        if (a == "--same-device") { o.same_device = true; continue; }
        if (a == "--strict") { o.strict = true; continue; }
        if (a.rfind("--", 0) == 0) {
            std::string name = a.substr(2);
            size_t eq = name.find('=');
            bool has_val = eq != std::string::npos;
            if (has_val) { val = name.substr(eq + 1); name = name.substr(0, eq); }
            for (auto &op : opts) if (name == op.l) key = op.s;
            if (!key) die("unknown option " + a);
            if (!has_val) { if (i + 1 >= argc) die(a + " needs a value"); val = argv[++i]; }
        } else if (a.size() >= 2 && a[0] == '-') {
            for (auto &op : opts) if (a[1] == op.s) key = op.s;
            if (!key) die("unknown option " + a);
            if (a.size() > 2) val = a.substr(2);
            else { if (i + 1 >= argc) die(a + " needs a value"); val = argv[++i]; }
        } else {
            die("unexpected argument " + a);
        }
        bool ok = true;
        switch (key) {
        case 'f': o.dataset_path = val; break;
        case 'd': o.data_graph = val; break;
        case 'q': o.query_graph = val; break;
        case 'm': o.mode = val; break;
        case 'n': o.answers = val; break;
        case 'p': ok = parse_u32(val, &o.partition_num); break;
        case 'l': ok = parse_u32(val, &o.path_length); break;
        case 'e': ok = parse_u32(val, &o.vde_dim); break;
        }
        if (!ok) die("bad value for -" + std::string(1, key) + ": " + val);
    }
    return o;
}

// The reference's block file seeks with 32-bit offsets (`fseek(fp, (bnum - act_block) * blocklength, SEEK_CUR)`,
// include/blockfile/blk_file.h:33): an index.dat of 2 GiB or more -- its own insert-built ones included -- breaks its
// online run.  The file is still written (other consumers may read it); the user is told to use more partitions.
inline void warn_if_index_too_large_for_reference(const std::string &path)
{
    struct stat st;
    if (stat(path.c_str(), &st) == 0 && (uint64_t)st.st_size >= (1ull << 31))
        fprintf(stderr, "%s: warning: %s is %.2f GiB; the reference's online binary cannot seek in index files of 2 GiB or more "
                        "(blk_file.h:33) -- use a larger -p\n", g_tool, path.c_str(), st.st_size / 1073741824.0);
}

// directory of the running binary, with the trailing slash ("" when /proc/self/exe cannot be read): where libgnnpe_online.so lives
inline std::string exe_dir()
{
    char buf[4096];
    const ssize_t k = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
    if (k <= 0) return "";
    buf[k] = 0;
    std::string s(buf);
    const size_t cut = s.rfind('/');
    return cut == std::string::npos ? "" : s.substr(0, cut + 1);
}

// Size of the index.dat a partition of `points` paths becomes, so that the 2 GiB limit above is checked BEFORE anything is
// written: the library's own figure for the builder that will write the file (gnnpe_index_file_bytes: 0 = the pair-major build
// of the single-GPU path, 1 = the tuple-array build of --gpus N; their nodes hold different numbers of entries).
inline uint64_t index_file_bytes(uint64_t points, uint32_t D, int builder) { return gnnpe_index_file_bytes(points, D, builder); }
// --index: the message for a partition whose index.dat would reach 2 GiB (empty when every file stays below), naming the
// smallest -p that keeps partitions of equal size below it.  The caller refuses unless --allow-large.
inline std::string index_size_problem(const std::vector<uint64_t> &part_count, uint64_t P, uint32_t D, int builder)
{
    uint64_t worst = 0;
    uint32_t worst_pid = 0;
    if (index_file_bytes(1, D, builder) == 0)  // no node holds three entries of this width (rtnode.cpp:27-28): nothing to size
        die("--index: an entry of " + std::to_string(D) + " dimensions (" + std::to_string(16ull * D + 4) +
            " bytes) gives a node capacity below 3 in a 4096-byte block; use a smaller -e");
    for (uint32_t i = 0; i < part_count.size(); i++) {
        const uint64_t b = index_file_bytes(part_count[i], D, builder);
        if (b > worst) worst = b, worst_pid = i;
    }
    if (worst < (1ull << 31)) return "";
    uint32_t p_min = (uint32_t)part_count.size();
    while (index_file_bytes((P + p_min - 1) / p_min, D, builder) >= (1ull << 31)) p_min++;
    return "partition " + std::to_string(worst_pid) + " holds " + std::to_string(part_count[worst_pid]) + " paths: its index.dat would be " +
           std::to_string(worst >> 20) + " MiB, and the reference's online binary cannot seek in index files of 2 GiB or more "
           "(blk_file.h:33); partitions of equal size stay below that from -p " + std::to_string(p_min);
}

// R0 onto a context: the rows the enumeration runs on and, for a file that is not a simple graph, the rows as the reference's
// loader holds them (graph_loader.h) for gen_vde and the degree columns
inline int load_graph_into(gnnpe_ctx *ctx, const gnnpe_host::StaticGraph &g)
{
    int rc = gnnpe_load_csr(ctx, g.n, g.enum_offsets().data(), g.enum_neighbors().data(), g.labels.data());
    if (rc != 0 || g.simple) return rc;
    std::vector<uint64_t> off(g.offsets.begin(), g.offsets.end());
    return gnnpe_set_multigraph_rows(ctx, g.n, off.data(), g.neighbors.data());
}

inline bool is_dir(const std::string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

}  // namespace cli
