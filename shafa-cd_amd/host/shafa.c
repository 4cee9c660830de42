/*
 * shafa.c — command line of the MI355X Shafa codec.  Same grammar, module sequencing, conflict
 * rules, messages and exit codes as the reference CLI (reference src/shafa.c:47-317):
 *
 *     shafa <file> [-m f|t|c|d]... [-b K|m|M] [-c r|f] [-d s|r] [--no-multithread]
 *
 * The modules themselves (host/modules.c) run their per-block work on the GPU via libshafa_hip.so.
 */
#include "shafa_host.h"

#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct {
    unsigned long block_size;
    bool f, t, c, d;             /* -m */
    bool force_rle, force_freq;  /* -c r|f */
    bool d_shaf, d_rle;          /* -d s|r */
} options;

/* shafa.c:47-132: one positional file; every "-x" takes exactly one one-character value */
static bool parse_args(int argc, char *const argv[], options *o, const char **file)
{
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        if (!strcmp(a, "--no-multithread")) { NO_MULTITHREAD = true; continue; }
        if (a[0] != '-') {
            if (*file) return false;
            *file = a;
            continue;
        }
        if (++i >= argc) return false;
        const char *v = argv[i];
        if (strlen(a) != 2 || strlen(v) != 1) return false;
        switch (a[1]) {
        case 'm':
            if (*v == 'f') o->f = true; else if (*v == 't') o->t = true;
            else if (*v == 'c') o->c = true; else if (*v == 'd') o->d = true;
            else return false;
            break;
        case 'b':
            if (*v == 'K') o->block_size = SHAFA_640KiB; else if (*v == 'm') o->block_size = SHAFA_8MiB;
            else if (*v == 'M') o->block_size = SHAFA_64MiB;
            else return false;
            break;
        case 'c':
            if (*v == 'r') o->force_rle = true; else if (*v == 'f') o->force_freq = true;
            else return false;
            break;
        case 'd':
            if (*v == 's') o->d_shaf = true; else if (*v == 'r') o->d_rle = true;
            else return false;
            break;
        default:
            return false;
        }
    }
    return true;
}

/* shafa.c:150-259 */
static int run_modules(const options *o, char **file)
{
    int err;
    /* f, t and c in one process (the default run, shafa.c:293-298): one upload per block where the blocks are large;
       SHAFA_FTC=0 keeps the three separate passes */
    bool ftc_done = false;
    if (o->f && o->t && o->c) {
        const char *e = getenv("SHAFA_FTC");
        if (!(e && *e == '0')) ftc_done = shafa_ftc_compress(file, o->force_rle, o->force_freq, o->block_size) == SHAFA_SUCCESS;
        if (e && *e == '2' && !ftc_done) {             /* (tests: the one-upload driver must have done the run) */
            fputs("SHAFA_FTC=2: the F -> T -> C driver did not take this run\n", stderr);
            return SHAFA_OUTSIDE_MODULE;
        }
    }
    if (o->f && !ftc_done) {
        err = freq_rle_compress(file, o->force_rle, o->force_freq, o->block_size);
        if (err) {
            fputs("Module f: Something went wrong while compressing with RLE or creating frequencies' table...\n", stderr);
            return err;
        }
    }
    if (o->t && !ftc_done) {
        if (!o->f) {                       /* -m t alone works on X.freq and strips the suffix */
            if (!shafa_has_ext(*file, SHAFA_FREQ_EXT)) {
                fprintf(stderr, "Module t: Wrong extension... Should end in %s\n", SHAFA_FREQ_EXT);
                return SHAFA_OUTSIDE_MODULE;
            }
            char *stem = shafa_rm_ext(*file);
            if (!stem) return SHAFA_LACK_OF_MEMORY;
            free(*file);
            *file = stem;
        }
        err = get_shafa_codes(*file);
        if (err) { fputs("Module t: Something went wrong...\n", stderr); return err; }
    }
    if (o->c && !ftc_done) {
        if (o->f && !o->t) {
            fputs("Module c: Can't execute module 'c' after 'f' without 't'...\n", stderr);
            return SHAFA_OUTSIDE_MODULE;
        }
        err = shafa_compress(file);
        if (err) { fputs("Module c: Something went wrong...\n", stderr); return err; }
    }
    if (o->d) {
        if ((o->f && (!o->t || !o->c) && !shafa_has_ext(*file, SHAFA_RLE_EXT)) || (o->t && !o->c)) {
            fputs("Module d: Can't execute module 'd' after 'f' without 't' or 'c', nor execute it after 't'  without 'c'...\n", stderr);
            return SHAFA_OUTSIDE_MODULE;
        }
        bool done = false;
        if (o->d_shaf || !o->d_rle) {
            if (!shafa_has_ext(*file, SHAFA_SHAFA_EXT)) {
                if (o->d_shaf) {
                    fprintf(stderr, "Module d: Wrong extension... Should end in %s\n", SHAFA_SHAFA_EXT);
                    return SHAFA_OUTSIDE_MODULE;
                }
            } else {
                const bool is_rle_shaf = shafa_has_ext(*file, SHAFA_RLE_EXT SHAFA_SHAFA_EXT);
                if (o->d_rle && !is_rle_shaf) {
                    fprintf(stderr, "Module d: Wrong extension... Should end in %s\n", SHAFA_RLE_EXT SHAFA_SHAFA_EXT);
                    return SHAFA_OUTSIDE_MODULE;
                }
                err = shafa_decompress(file, (o->d_rle || !o->d_shaf) && is_rle_shaf);
                if (err) { fputs("Module d: Something went wrong while decompressing...\n", stderr); return err; }
                done = true;
            }
        }
        if (!done && (o->d_rle || !o->d_shaf)) {
            if (!shafa_has_ext(*file, SHAFA_RLE_EXT)) {
                fprintf(stderr, "Module d: Wrong extension... Should end in %s\n", SHAFA_RLE_EXT);
                return SHAFA_OUTSIDE_MODULE;
            }
            err = rle_decompress(file);
            if (err) { fputs("Module d: Something went wrong while decompressing...\n", stderr); return err; }
        }
    }
    return SHAFA_SUCCESS;
}

int main(int argc, char *const argv[])
{
    options o;
    memset(&o, 0, sizeof(o));
    const char *arg_file = NULL;
    if (argc <= 1) { fputs("No file input\n", stderr); return 1; }
    if (!parse_args(argc, argv, &o, &arg_file)) { fputs("Wrong Options' syntax\n", stderr); return 1; }
    if (!arg_file) { fputs("No file input\n", stderr); return 1; }
    char *file = shafa_add_ext(arg_file, "");
    if (!file) { fputs("Not enough memory\n", stderr); return 1; }

    if (!o.f && !o.t && !o.c && !o.d) {            /* shafa.c:293-298 */
        if (shafa_has_ext(file, SHAFA_SHAFA_EXT)) o.d = true;
        else o.f = o.t = o.c = true;
    }
    if (!o.block_size) o.block_size = SHAFA_64KiB;  /* shafa.c:304-305 */

    /* The reference's -m f/c/d start one thread per block on the host's cores; here the blocks of a file go to every
     * GPU of the node (three in flight per GPU, retired in order).  SHAFA_DEVICES="0,2,3" restricts the set; a list that
     * names a GPU the node does not have is an error, not a silent fall-back to device 0.  Module T alone runs on the host
     * and touches no GPU. */
    if (o.f || o.c || o.d) {
        int devs[64], nd = 0;
        const char *e = getenv("SHAFA_DEVICES");
        bool bad = false;
        for (const char *q = e; q && *q && nd < 64;) {
            char *end = NULL;
            const long v = strtol(q, &end, 10);
            if (end == q || v < 0 || v > 1023) { bad = true; break; }
            devs[nd++] = (int)v;
            if (*end && *end != ',') { bad = true; break; }
            q = (*end == ',') ? end + 1 : end;
        }
        if (e && *e && (bad || nd == 0 || shafa_hip_init_devices(devs, nd) != SHAFA_SUCCESS)) {
            fprintf(stderr, "SHAFA_DEVICES=%s does not name GPUs of this node (%d visible)\n", e, shafa_hip_device_count());
            free(file);
            return 1;
        }
        if (!(e && *e) && shafa_hip_device_count() > 1) (void)shafa_hip_init_devices(NULL, 0);   /* every visible GPU */
    }
    const int err = run_modules(&o, &file);
    free(file);
    if (err && err != SHAFA_OUTSIDE_MODULE) fputs(shafa_error_msg(err), stderr);
    /* Every output file is written and closed by now.  Leave without tearing the GPU context down buffer by buffer (the
     * kept pipe's pinned memory, the runtime's exit handlers: 0.1-0.15 s of a 0.7 s run on a 2 GiB file): the process ends
     * here and the kernel driver reclaims all of it.  SHAFA_CLEAN_EXIT=1 takes the long way (leak checkers). */
    fflush(NULL);
    if (getenv("SHAFA_CLEAN_EXIT")) {
        shafa_host_release();
        shafa_hip_shutdown();
        return err ? 1 : 0;
    }
    _exit(err ? 1 : 0);
}
