/* main.c -- `fermi-amd`: the sub-commands of fermi (main.c:101-124) that sit on the FMD hot path,
 * same argv surface as cmd.c, index work on MI355X.  No CPU fallback: without a GPU it says so. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include "fmd_host.h"

static int main_unitig(int argc, char *argv[]) /* cmd.c:184-216 */
{
    int c, min_match = 30, device = 0;
    while ((c = getopt(argc, argv, "Ml:t:r:g:")) >= 0) {
        switch (c) {
        case 'l': min_match = atoi(optarg); break;
        case 'M': break;                 /* mmap: meaningless for a device-resident index */
        case 't': break;                 /* threads: the walk is the deterministic -t1 walk */
        case 'g': device = atoi(optarg); break;
        case 'r': fprintf(stderr, "[E::%s] -r (rank file) is not supported yet\n", __func__); return 1;
        }
    }
    if (optind + 1 > argc) {
        fprintf(stderr, "\nUsage:   fermi-amd unitig [options] <reads.fmd>\n\n");
        fprintf(stderr, "Options: -l INT      min match [%d]\n", min_match);
        fprintf(stderr, "         -t INT      number of threads [ignored: output is that of -t1]\n");
        fprintf(stderr, "         -g INT      GPU to use [0]\n\n");
        return 1;
    }
    return fmdh_unitig(argv[optind], device, min_match, stdout);
}

int main(int argc, char *argv[])
{
    if (argc < 2) {
        fprintf(stderr, "\nProgram: fermi-amd (FMD-index hot path of fermi on AMD MI355X)\n\n");
        fprintf(stderr, "Usage:   fermi-amd <command> [arguments]\n\n");
        fprintf(stderr, "Command: unitig     construct unitigs (fermi unitig)\n\n");
        return 1;
    }
    if (fmd_device_count() <= 0) {
        fprintf(stderr, "[E::main] %s\n", fmd_strerror(FMD_E_NODEV));
        return 1;
    }
    if (strcmp(argv[1], "unitig") == 0) return main_unitig(argc - 1, argv + 1);
    fprintf(stderr, "[E::main] unrecognized command `%s'\n", argv[1]);
    return 1;
}
