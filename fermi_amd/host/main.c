#define _GNU_SOURCE
/* main.c -- `fermi-amd`: the sub-commands of fermi (main.c:101-124) that sit on the FMD hot path,
 * same argv surface as cmd.c, index work on MI355X.  No CPU fallback: without a GPU it says so. */
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <unistd.h>
#include "fmd_host.h"

/* "-g 0,1,3": one entry per replica of the index (a GPU may be listed twice) */
static int parse_gpu_list(const char *s, int *dev, int max)
{
    int n = 0;
    while (*s && n < max) {
        char *e;
        long v = strtol(s, &e, 10);
        if (e == s || v < 0) return -1;
        dev[n++] = (int)v;
        s = *e == ',' ? e + 1 : e;
        if (*e && *e != ',') return -1;
    }
    return n;
}

static int main_unitig(int argc, char *argv[]) /* cmd.c:184-216 */
{
    int c, min_match = 30, n_dev = 1, devices[64] = {0};
    const char *rank_file = 0;
    while ((c = getopt(argc, argv, "Ml:t:r:g:")) >= 0) {
        switch (c) {
        case 'l': min_match = atoi(optarg); break;
        case 'M': break;                 /* mmap: meaningless for a device-resident index */
        case 't': if (atoi(optarg) > 0) setenv("FMD_WALK_THREADS", optarg, 0); break;   /* host threads of the walk (unitig_walk.c); whatever the number, the MAG is -t1's */
        case 'g': n_dev = parse_gpu_list(optarg, devices, 64); break;
        case 'r': rank_file = optarg; break;
        }
    }
    if (n_dev < 1) { fprintf(stderr, "[E::%s] -g takes a comma-separated list of GPU numbers\n", __func__); return 1; }
    for (c = 0; c < n_dev; ++c)
        if (devices[c] >= fmd_device_count()) { fprintf(stderr, "[E::%s] GPU %d: this node has %d\n", __func__, devices[c], fmd_device_count()); return 1; }
    if (optind + 1 > argc) {
        fprintf(stderr, "\nUsage:   fermi-amd unitig [options] <reads.fmd>\n\n");
        fprintf(stderr, "Options: -l INT      min match [%d]\n", min_match);
        fprintf(stderr, "         -t INT      host threads of the walk; the output is that of `fermi unitig -t1` whatever the number [16]\n");
        fprintf(stderr, "         -r FILE     rank file [null]\n");
        fprintf(stderr, "         -g LIST     GPUs to use, e.g. 0,1,2,3: the index is replicated on each, GPU g computes the\n");
        fprintf(stderr, "                     sequences i = g (mod #GPUs) as worker g of `fermi unitig -t` would seed them [0]\n\n");
        return 1;
    }
    return fmdh_unitig(argv[optind], n_dev, devices, min_match, rank_file, stdout);
}

static int main_seqsort(int argc, char *argv[]) /* cmd.c:486-505 */
{
    int c, device = 0;
    uint64_t *sorted = 0, n = 0;
    while ((c = getopt(argc, argv, "t:g:")) >= 0) if (c == 'g') device = atoi(optarg);
    if (optind == argc) { fprintf(stderr, "Usage: fermi-amd seqsort [-g GPU] <reads.fmd>\n"); return 1; }
    if (fmdh_seqsort(argv[optind], device, &sorted, &n)) return 1;
    fwrite(sorted, 8, n, stdout);
    free(sorted);
    return 0;
}

/* "-g 0,1,3" -> devices[]; returns the count (1 device, 0, when the option is absent) */
static int parse_gpus(const char *arg, int *devices, int max)
{
    int n = 0;
    const char *p = arg;
    while (*p && n < max) {
        char *q;
        const long v = strtol(p, &q, 10);
        if (q == p) break;
        devices[n++] = (int)v;
        p = *q == ',' ? q + 1 : q;
        if (*q != ',') break;
    }
    if (n == 0) { devices[0] = 0; n = 1; }
    return n;
}

static int main_build(int argc, char *argv[]) /* cmd.c:378-484 */
{
    int c, force = 0, max_len = 0x7fffffff, no_fr = 1, device = 0;
    const char *out = "-";
    while ((c = getopt(argc, argv, "fb:o:i:s:l:Og:")) >= 0) {
        switch (c) {
        case 'f': force = 1; break;
        case 'o': out = optarg; break;
        case 'l': max_len = atoi(optarg); break;
        case 'O': no_fr = 0; break;
        case 's': break;   /* symbols per SA-IS block: the GPU sorts everything at once */
        case 'g': device = atoi(optarg); break;
        case 'b': if (atoi(optarg) != 3) { fprintf(stderr, "[E::%s] only -b 3 is supported\n", __func__); return 1; } break;
        case 'i': fprintf(stderr, "[E::%s] -i (append to an index) is not supported\n", __func__); return 1;
        }
    }
    if (argc == optind) {
        fprintf(stderr, "\nUsage:   fermi-amd build [options] <in.fa>\n\n");
        fprintf(stderr, "Options: -f        force to overwrite the output file (effective with -o)\n");
        fprintf(stderr, "         -l INT    trim read down to INT bp [inf]\n");
        fprintf(stderr, "         -o FILE   output file name [stdout]\n");
        fprintf(stderr, "         -O        do not trim 1bp for reads whose forward and reverse are identical\n");
        fprintf(stderr, "         -g INT    GPU to use [0]\n\n");
        return 1;
    }
    if (strcmp(out, "-") && !force) {
        FILE *fp = fopen(out, "rb");
        if (fp) { fclose(fp); fprintf(stderr, "[E::%s] File `%s' exists. Please use `-f' to overwrite.\n", __func__, out); return 1; }
    }
    return fmdh_build(argv[optind], out, device, max_len, no_fr);
}

static int main_exact(int argc, char *argv[]) /* cmd.c:292-331 */
{
    int c, self_match = 0, devices[FMDH_MAX_GPUS] = {0}, n_dev = 1;
    while ((c = getopt(argc, argv, "Msg:")) >= 0) {
        switch (c) {
        case 'M': break;
        case 's': self_match = 1; break;
        case 'g': n_dev = parse_gpus(optarg, devices, FMDH_MAX_GPUS); break;
        }
    }
    if (optind + 2 > argc) { fprintf(stderr, "Usage: fermi-amd exact [-s] [-g GPU[,GPU..]] <idxbase.fmd> <src.fa>\n"); return 1; }
    return fmdh_exact_multi(argv[optind], argv[optind + 1], n_dev, devices, self_match, stdout);
}

static int main_correct(int argc, char *argv[]) /* cmd.c:253-291 */
{
    int c, devices[FMDH_MAX_GPUS] = {0}, n_dev = 1;
    fmdh_ecopt_t opt;
    opt.w = -1; opt.min_occ = 3; opt.keep_bad = 0; opt.is_paired = 0; opt.max_corr = 0.3f; opt.trim_l = 0; opt.step = 5;
    while ((c = getopt(argc, argv, "MKt:k:v:O:pC:l:s:g:")) >= 0) {
        switch (c) {
        case 'M': case 'v': break;             /* mmap / verbosity: no effect on the output */
        case 't': fmdh_correct_set_threads(atoi(optarg)); break; /* ec_fix workers, as correct.c:281-290 */
        case 'K': opt.keep_bad = 1; break;
        case 'k': opt.w = atoi(optarg); break;
        case 'O': opt.min_occ = atoi(optarg); break;
        case 'p': opt.is_paired = 1; break;
        case 'C': opt.max_corr = (float)atof(optarg); break;
        case 'l': opt.trim_l = atoi(optarg); break;
        case 's': opt.step = atoi(optarg); break;
        case 'g': n_dev = parse_gpus(optarg, devices, FMDH_MAX_GPUS); break;
        }
    }
    if (optind + 2 > argc) {
        fprintf(stderr, "\nUsage:   fermi-amd correct [options] <reads.fmd> <reads.fq>\n\n");
        fprintf(stderr, "Options: -k INT      k-mer length; -1 for auto [%d]\n", opt.w);
        fprintf(stderr, "         -O INT      minimum (k+1)-mer occurrences [%d]\n", opt.min_occ);
        fprintf(stderr, "         -C FLOAT    max fraction of corrected bases [%.2f]\n", opt.max_corr);
        fprintf(stderr, "         -l INT      trim read down to INT bp; 0 to disable [0]\n");
        fprintf(stderr, "         -s INT      step size for the jumping heuristic; 0 to disable [%d]\n", opt.step);
        fprintf(stderr, "         -t INT      number of host threads for the correction pass [1]\n");
        fprintf(stderr, "         -K          keep bad/unfixable reads\n");
        fprintf(stderr, "         -p          paired-end reads (interleaved)\n");
        fprintf(stderr, "         -g LIST     GPUs to use, e.g. 0,1,2,3: harvest by last base on up to four, reads of a batch split over all [0]\n\n");
        return 1;
    }
    return fmdh_correct_multi(argv[optind], argv[optind + 1], n_dev, devices, &opt, stdout);
}

static int main_chkbwt(int argc, char *argv[]) /* cmd.c:47-130 */
{
    int c, plain = 0, check_rank = 0, device = 0;
    while ((c = getopt(argc, argv, "pMrg:")) >= 0) {
        switch (c) {
        case 'p': plain = 1; break;
        case 'r': check_rank = 1; break;
        case 'M': break;
        case 'g': device = atoi(optarg); break;
        }
    }
    if (argc == optind) {
        fprintf(stderr, "\nUsage:   fermi-amd chkbwt [options] <idxbase.fmd>\n\n");
        fprintf(stderr, "Options: -r        check rank\n");
        fprintf(stderr, "         -p        print the BWT to the stdout\n");
        fprintf(stderr, "         -g INT    GPU to use [0]\n\n");
        return 1;
    }
    return fmdh_chkbwt(argv[optind], device, plain, check_rank, stdout);
}

static int main_unpack(int argc, char *argv[]) /* cmd.c:142-171 */
{
    int c, n = 0, m = 0, device = 0, rc;
    uint64_t *list = 0;
    while ((c = getopt(argc, argv, "Mi:g:")) >= 0) {
        switch (c) {
        case 'i':
            if (n == m) { m = m ? m << 1 : 16; list = (uint64_t *)realloc(list, 8 * (size_t)m); }
            list[n++] = (uint64_t)atol(optarg);
            break;
        case 'M': break;
        case 'g': device = atoi(optarg); break;
        }
    }
    if (argc == optind) {
        fprintf(stderr, "\nUsage:   fermi-amd unpack [-i index] [-g GPU] <seqs.fmd>\n\n");
        fprintf(stderr, "Options: -i INT    index of the read to output, starting from 0 [null]\n\n");
        return 1;
    }
    rc = fmdh_unpack(argv[optind], device, n, list, stdout);
    free(list);
    return rc;
}

static int main_remap(int argc, char *argv[]) /* cmd.c:218-251 */
{
    int c, device = 0;
    fmdh_remapopt_t opt;
    const char *rank_file = 0;
    opt.skip = 50; opt.min_pcv = 0; opt.max_dist = 1000;
    while ((c = getopt(argc, argv, "Ml:t:c:r:D:g:")) >= 0) {
        switch (c) {
        case 'l': opt.skip = atoi(optarg); break;
        case 'c': opt.min_pcv = atoi(optarg); break;
        case 'D': opt.max_dist = atoi(optarg); break;
        case 'r': rank_file = optarg; break;
        case 'M': case 't': break; /* mmap / threads: the output is that of -t1 (contigs in input order) */
        case 'g': device = atoi(optarg); break;
        }
    }
    if (optind + 2 > argc) {
        fprintf(stderr, "\nUsage:   fermi-amd remap [options] <reads.fmd> <contigs.fq>\n\n");
        fprintf(stderr, "Options: -l INT      skip ending INT bases of a read pair [%d]\n", opt.skip);
        fprintf(stderr, "         -c INT      minimum paired-end coverage [%d]\n", opt.min_pcv);
        fprintf(stderr, "         -D INT      maximum insert size (external distance) [%d]\n", opt.max_dist);
        fprintf(stderr, "         -r FILE     rank [null]\n");
        fprintf(stderr, "         -g INT      GPU to use [0]\n\n");
        return 1;
    }
    return fmdh_remap(argv[optind], argv[optind + 1], device, &opt, rank_file, stdout);
}

#include <time.h>
static double main_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
/* On a host with several NUMA nodes `unitig` stays on the node it started on (FMD_NUMA=off: wherever the scheduler puts it; FMD_NUMA=n: node n; FMD_NUMA=all:
 * every sub-command is pinned, not only `unitig`): its table is filled by the host threads and then read at random by the thread that commits -- first touch
 * puts a page on the toucher's node, and a walk that runs on the other one pays the remote latency on half of its steps.  The threads created afterwards,
 * the runtime's included, inherit the mask.  The new mask is the INHERITED one AND the node's CPUs: a binding the caller made (taskset, numactl, a batch
 * system's cpuset) is never widened, and a caller that already bound the process to fewer than the online CPUs has decided -- nothing is done then unless
 * FMD_NUMA names a node.  `correct -tN` and `build` keep every core they were given.  Returns the node, -1 when nothing was done. */
static int stay_on_one_node(const char *cmd)
{
    const char *e = getenv("FMD_NUMA");
    const int named = e && e[0] >= '0' && e[0] <= '9', all_cmds = e && strcmp(e, "all") == 0;
    int node, n_nodes = 0, cpu = sched_getcpu(), found = -1, i, n_inherited, n_new = 0;
    char path[128], buf[4096];
    cpu_set_t set, inherited;
    if (e && strcmp(e, "off") == 0) return -1;
    if (!named && !all_cmds && strcmp(cmd, "unitig") != 0) return -1;
    if (sched_getaffinity(0, sizeof(inherited), &inherited) != 0) return -1;
    n_inherited = CPU_COUNT(&inherited);
    if (!named && n_inherited < (int)sysconf(_SC_NPROCESSORS_ONLN)) return -1;   /* bound by the caller already */
    CPU_ZERO(&set);
    for (node = 0; node < 64; ++node) {
        FILE *f;
        char *q;
        cpu_set_t cs;
        int any = 0;
        snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
        f = fopen(path, "r");
        if (!f) break;
        ++n_nodes;
        if (!fgets(buf, sizeof(buf), f)) buf[0] = 0;
        fclose(f);
        CPU_ZERO(&cs);
        for (q = buf; *q && *q != '\n';) {          /* "0-63,128-191" */
            long a = strtol(q, &q, 10), b = a;
            if (*q == '-') b = strtol(q + 1, &q, 10);
            for (; a <= b && a < CPU_SETSIZE; ++a) { CPU_SET((int)a, &cs); any = 1; }
            if (*q == ',') ++q;
            else if (*q && *q != '\n') break;       /* not a cpulist */
        }
        if (any && ((named && atoi(e) == node) || (!named && cpu >= 0 && cpu < CPU_SETSIZE && CPU_ISSET(cpu, &cs)))) { set = cs; found = node; }
    }
    if (n_nodes < 2 || found < 0) return -1;
    for (i = 0; i < CPU_SETSIZE; ++i)
        if (CPU_ISSET(i, &set)) { if (CPU_ISSET(i, &inherited)) ++n_new; else CPU_CLR(i, &set); }
    if (n_new == 0 || n_new == n_inherited) return -1;      /* nothing of the node is ours / nothing would change */
    return sched_setaffinity(0, sizeof(set), &set) == 0 ? found : -1;
}

int main(int argc, char *argv[])
{
    if (argc < 2) {
        fprintf(stderr, "\nProgram: fermi-amd (FMD-index hot path of fermi on AMD MI355X)\n\n");
        fprintf(stderr, "Usage:   fermi-amd <command> [arguments]\n\n");
        fprintf(stderr, "Command: build      generate the FMD-index (fermi build)\n");
        fprintf(stderr, "         seqsort    rank -> read index map for `unitig -r` (fermi seqsort)\n");
        fprintf(stderr, "         unitig     construct unitigs (fermi unitig)\n");
        fprintf(stderr, "         correct    error correction (fermi correct)\n");
        fprintf(stderr, "         exact      find super-maximal exact matches (fermi exact)\n");
        fprintf(stderr, "         chkbwt     print / check the BWT held on the GPU (fermi chkbwt)\n");
        fprintf(stderr, "         unpack     print the indexed sequences (fermi unpack)\n");
        fprintf(stderr, "         remap      coverage of contigs by the reads, paired-end breaks (fermi remap)\n\n");
        fprintf(stderr, "Environment: FMD_NUMA=off|all|<node>  `unitig` keeps to the CPUs of the NUMA node it started on (never more than the\n");
        fprintf(stderr, "                                      mask it inherited; nothing is done under taskset / numactl / a cpuset);\n");
        fprintf(stderr, "                                      off = no pinning, all = every command, <node> = that node\n\n");
        return 1;
    }
    const double t_start = main_now();
    const int timing = getenv("FMD_TIMING") != 0;
    int rc;
    setvbuf(stdout, 0, _IOFBF, 4 << 20); /* the outputs are hundreds of MB of short lines */
    { const int node = stay_on_one_node(argv[1]); if (timing && node >= 0) fprintf(stderr, "[M::main] the process stays on NUMA node %d\n", node); }
    if (fmd_device_count() <= 0) {
        fprintf(stderr, "[E::main] %s\n", fmd_strerror(FMD_E_NODEV));
        return 1;
    }
    if (timing) fprintf(stderr, "[M::main] the runtime is up: %.3f s\n", main_now() - t_start);
    if (strcmp(argv[1], "unitig") == 0) rc = main_unitig(argc - 1, argv + 1);
    else if (strcmp(argv[1], "build") == 0) rc = main_build(argc - 1, argv + 1);
    else if (strcmp(argv[1], "seqsort") == 0) rc = main_seqsort(argc - 1, argv + 1);
    else if (strcmp(argv[1], "exact") == 0) rc = main_exact(argc - 1, argv + 1);
    else if (strcmp(argv[1], "correct") == 0) rc = main_correct(argc - 1, argv + 1);
    else if (strcmp(argv[1], "remap") == 0) rc = main_remap(argc - 1, argv + 1);
    else if (strcmp(argv[1], "chkbwt") == 0) rc = main_chkbwt(argc - 1, argv + 1);
    else if (strcmp(argv[1], "unpack") == 0) rc = main_unpack(argc - 1, argv + 1);
    else { fprintf(stderr, "[E::main] unrecognized command `%s'\n", argv[1]); return 1; }
    /* (what a caller's clock sees beyond this: 0.6 s at 10^7 reads for loading the runtime's libraries before main and for the kernel taking the process's
     * mappings and its GPU context down after it -- ending the process with _exit instead of the exit handlers changed nothing measurable) */
    if (timing) {
        fflush(stdout);
        fprintf(stderr, "[M::main] %s: %.3f s from the start of the process\n", argv[1], main_now() - t_start);
        fprintf(stderr, "[M::main] %s: peak resident set %.2f GB\n", argv[1], fmdh_rss_gb(1));   /* (VmHWM: of this program, not of what exec'ed it) */
    }
    return rc;
}
