// afq_cli.cpp — `afquant quant …` and `afquant infer …`: the flag surfaces of `alevin-fry quant` (src/main.rs:294-348 of the
// reference) and `alevin-fry infer` (src/main.rs:350-365) in front of afq_quantify() / afq_infer_files()
// (include/afquant_host.h).  Same spellings and defaults; what the reference refuses (--use-eds, -b with a plain
// resolution, -d with trivial, --summary-stat without -b) is refused here too, with its message.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/afquant.h"
#include "../../include/afquant_host.h"

static void usage() {
    std::fprintf(stderr,
                 "usage: afquant quant -i <input-dir> -m <tg-map> -o <output-dir> -r <resolution>\n"
                 "       [-t <threads>] [--small-thresh N] [--umi-edit-dist 0|1] [--large-graph-thresh N]\n"
                 "       [--quant-subset FILE] [--init-uniform] [--use-mtx] [-d] [-b N] [--device N | --devices 0,1,...]\n"
                 "       [--summary-stat] [--boot-seed S] [--sa-model winner-take-all|prefer-ambig]\n"
                 "resolutions: trivial cr-like cr-like-em parsimony parsimony-em parsimony-gene parsimony-gene-em\n"
                 "       afquant atac deduplicate -i <input-dir> [-t N] [-d fw|rc] [--device N]\n"
                 "       afquant infer -c <geqc_counts.mtx> -e <gene_eqclass.txt.gz> -o <output-dir> [--usa] [--quant-subset FILE] [-t N]\n");
}

int main(int argc, char** argv) {
    // The rows of a range come back over PCIe while the next range's kernels run: that copy belongs on the DMA engines.  Left to
    // itself the HIP runtime moves some of these copies with blit kernels on the compute queue, where they take a millisecond of
    // the next decode (profiles/r04_timeline_*.txt); this keeps every copy on SDMA.  Read by the runtime at its first call.
    setenv("GPU_FORCE_BLIT_COPY_SIZE", "0", 0);
    if (argc >= 2 && (std::strcmp(argv[1], "-h") == 0 || std::strcmp(argv[1], "--help") == 0)) { usage(); return 0; }
    if (argc >= 2 && (std::strcmp(argv[1], "-V") == 0 || std::strcmp(argv[1], "--version") == 0)) {
        std::printf("afquant-hip 0.1 (alevin-fry 0.18.0 quant / infer semantics, C ABI version %d)\n", AFQ_ABI_VERSION);
        return 0;
    }
    if (argc >= 2 && std::strcmp(argv[1], "infer") == 0) {   // src/main.rs:350-365, 825-847
        afq_infer_opts io{};
        auto need2 = [&](int& i) -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "-c" || a == "--count-mat") io.count_mat = need2(i);
            else if (a == "-e" || a == "--eq-labels") io.eq_labels = need2(i);
            else if (a == "-o" || a == "--output-dir") io.output_dir = need2(i);
            else if (a == "-t" || a == "--threads") io.num_threads = (uint32_t)std::atoi(need2(i));
            else if (a == "--usa") io.usa_mode = 1;
            else if (a == "--quant-subset") io.filter_list = need2(i);
            else if (a == "--use-mtx") {}
            else if (a == "--use-eds") { std::fprintf(stderr, "--use-eds is no longer supported. EDS output has been removed as of v0.12.\n"); return 1; }
            else if (a == "--device") io.device = (uint32_t)std::atoi(need2(i));
            else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
        }
        if (!io.count_mat || !io.eq_labels || !io.output_dir) { usage(); return 2; }
        const int rc = afq_infer_files(&io);
        if (rc) { std::fprintf(stderr, "afquant infer failed (%d): %s\n", rc, afq_host_last_error()); return 1; }
        return 0;
    }
    if (argc >= 3 && std::strcmp(argv[1], "atac") == 0 && std::strcmp(argv[2], "deduplicate") == 0) {   // src/main.rs:942-956, atac/run.rs:128-169
        afq_atac_dedup_opts ao{};
        ao.rev = 1;   // --permit-bc-ori defaults to rc
        auto need3 = [&](int& i) -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        for (int i = 3; i < argc; ++i) {
            const std::string a = argv[i];
            if (a == "-i" || a == "--input-dir") ao.input_dir = need3(i);
            else if (a == "-t" || a == "--threads") ao.num_threads = (uint32_t)std::atoi(need3(i));
            else if (a == "-d" || a == "--permit-bc-ori") {
                std::string v = need3(i);
                for (auto& ch : v) ch = (char)std::toupper((unsigned char)ch);
                if (v == "RC") ao.rev = 1; else if (v == "FW") ao.rev = 0;
                else { std::fprintf(stderr, "invalid barcode orientation %s\n", v.c_str()); return 2; }
            }
            else if (a == "--device") ao.device = (uint32_t)std::atoi(need3(i));
            else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
        }
        if (!ao.input_dir) { usage(); return 2; }
        const int rc = afq_atac_deduplicate(&ao);
        if (rc) { std::fprintf(stderr, "afquant atac deduplicate failed (%d): %s\n", rc, afq_host_last_error()); return 1; }
        return 0;
    }
    if (argc < 2 || std::strcmp(argv[1], "quant") != 0) { usage(); return 2; }
    afq_quant_opts o{};
    o.small_thresh = 100; o.umi_edit_dist = -1; o.large_graph_thresh = -1; o.num_threads = 0;
    std::string cmdline;
    std::vector<int32_t> devs;
    for (int i = 0; i < argc; ++i) { if (i) cmdline += ' '; cmdline += argv[i]; }
    o.cmdline = cmdline.c_str();
    auto need = [&](int& i) -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-i" || a == "--input-dir") o.input_dir = need(i);
        else if (a == "-m" || a == "--tg-map") o.tg_map = need(i);
        else if (a == "-o" || a == "--output-dir") o.output_dir = need(i);
        else if (a == "-r" || a == "--resolution") o.resolution = need(i);
        else if (a == "-t" || a == "--threads") o.num_threads = (uint32_t)std::atoi(need(i));
        else if (a == "--small-thresh") o.small_thresh = (uint32_t)std::atoi(need(i));
        else if (a == "--umi-edit-dist") o.umi_edit_dist = std::atoi(need(i));
        else if (a == "--large-graph-thresh") o.large_graph_thresh = std::atoi(need(i));
        else if (a == "--quant-subset") o.filter_list = need(i);
        else if (a == "--init-uniform") o.init_uniform = 1;
        else if (a == "--summary-stat") o.summary_stat = 1;
        else if (a == "--boot-seed") o.boot_seed = std::strtoull(need(i), nullptr, 10);
        else if (a == "--use-mtx") {}
        else if (a == "--use-eds") { std::fprintf(stderr, "--use-eds is no longer supported. EDS output has been removed as of v0.12.\n"); return 1; }
        else if (a == "-d" || a == "--dump-eqclasses") o.dump_eq = 1;
        else if (a == "-b" || a == "--num-bootstraps") o.num_bootstraps = (uint32_t)std::atoi(need(i));
        else if (a == "--sa-model") {
            const std::string v = need(i);
            if (v == "winner-take-all") o.sa_model = AFQ_SA_WINNER_TAKE_ALL;
            else if (v == "prefer-ambig") o.sa_model = AFQ_SA_PREFER_AMBIG;
            else { std::fprintf(stderr, "invalid value '%s' for '--sa-model': possible values: prefer-ambig, winner-take-all\n", v.c_str()); return 2; }
        }
        else if (a == "--multi-sample-output") (void)need(i);
        else if (a == "--device") o.device = (uint32_t)std::atoi(need(i));
        else if (a == "--devices") {   // comma list of HIP device ordinals: cells are range-partitioned over them
            for (const char* p = need(i); *p;) { char* e; const long v = std::strtol(p, &e, 10); if (e == p) { std::fprintf(stderr, "--devices wants a comma list of integers\n"); return 2; } devs.push_back((int32_t)v); p = *e == ',' ? e + 1 : e; if (*e && *e != ',') { std::fprintf(stderr, "--devices wants a comma list of integers\n"); return 2; } }
        }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
    }
    if (!devs.empty()) { o.devices = devs.data(); o.n_devices = (uint32_t)devs.size(); }
    if (!o.input_dir || !o.tg_map || !o.output_dir || !o.resolution) { usage(); return 2; }
    if ((o.summary_stat || o.init_uniform) && !o.num_bootstraps) {   // clap `requires("num-bootstraps")`, main.rs:306-307
        std::fprintf(stderr, "error: the following required arguments were not provided:\n  --num-bootstraps <NUMBOOTSTRAPS>\n");
        return 2;
    }
    const int rc = afq_quantify(&o);
    if (rc) { std::fprintf(stderr, "afquant quant failed (%d): %s\n", rc, afq_host_last_error()); return 1; }
    // every output file is written and closed: leave without unwinding the HIP runtime and the pinned pools (tens of ms of
    // teardown the operating system does anyway)
    std::fflush(nullptr);
    std::_Exit(0);
}
