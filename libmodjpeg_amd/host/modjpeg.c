/*
 * modjpeg — command line front end of the MI355X drop-in, option-compatible with the reference
 * CLI (reference src/contrib/modjpeg.c:32-173): the options are ACTIONS executed in command-line
 * order (`-x -d logo.png` pixelates, then composes; `-d logo.png -x` pixelates the logo too), the
 * write options -O/-P/-A apply to every later -o, -p/-m set state for later -d.
 *
 * Batch extension: the same action script applied to many files,
 *
 *     modjpeg --batch list.txt --outdir out/ [--jobs N] [--devices G]  -p br -m -10,-10 -d logo.png -y 20 -O
 *
 * where list.txt names one input JPEG per line ("-" = stdin).  Every file is read, run through
 * the script (all options except -i/-o, in order) and written to <outdir>/<basename>.  N host
 * threads work on different files; on a host with several GPUs thread t works on device t mod G
 * (G = --devices, default: every device present), each image from read to write on one device,
 * nothing crossing between devices.  An image is decoded on the GPU and STAYS there (shadow.c):
 * compose and the effects run in place, the encoder reads the planes where they are, and only
 * compressed bytes cross PCIe in either direction; concurrent decodes are launched as one batch.
 * Files the GPU entropy coders decline (progressive, arithmetic, -P/-A output) go through libjpeg
 * as in the reference.  The prepared dropon tiles are shared by all threads
 * (device-resident tile cache).
 */
#include <libmodjpeg.h>

#include <getopt.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

enum { OPT_BATCH = 1000, OPT_OUTDIR, OPT_JOBS, OPT_DEVICES };

static const struct option long_options[] = {
    {"input", required_argument, NULL, 'i'},     {"output", required_argument, NULL, 'o'},   {"dropon", required_argument, NULL, 'd'},
    {"position", required_argument, NULL, 'p'},  {"offset", required_argument, NULL, 'm'},   {"luminance", required_argument, NULL, 'y'},
    {"tintblue", required_argument, NULL, 'b'},  {"tintred", required_argument, NULL, 'r'},  {"pixelate", no_argument, NULL, 'x'},
    {"grayscale", no_argument, NULL, 'g'},       {"progressive", no_argument, NULL, 'P'},    {"optimize", no_argument, NULL, 'O'},
    {"arithmetric", no_argument, NULL, 'A'},     {"help", no_argument, NULL, 'h'},           {"batch", required_argument, NULL, OPT_BATCH},
    {"outdir", required_argument, NULL, OPT_OUTDIR}, {"jobs", required_argument, NULL, OPT_JOBS}, {"devices", required_argument, NULL, OPT_DEVICES},
    {NULL, 0, NULL, 0}};

typedef struct {
    int   code;
    char *arg;
} action_t;

/* interpreter state: what the reference keeps in main()'s locals */
typedef struct {
    mj_jpeg_t   image;
    mj_dropon_t dropon;
    unsigned    position;
    int         offset_x, offset_y, write_options;
} state_t;

static void state_init(state_t *s) {
    mj_init_jpeg(&s->image);
    mj_init_dropon(&s->dropon);
    s->position = MJ_ALIGN_TOP | MJ_ALIGN_LEFT; /* modjpeg.c:53 */
    s->offset_x = s->offset_y = s->write_options = 0;
}

static void state_free(state_t *s) {
    mj_free_jpeg(&s->image);
    mj_free_dropon(&s->dropon);
}

static void usage(void) {
    fputs("modjpeg (MI355X build) - options are executed in the order given\n\n"
          "  -i, --input file          read the JPEG to modify\n"
          "  -o, --output file         write the current image\n"
          "  -d, --dropon file[,mask]  compose a dropon (JPEG with optional mask JPEG, or PNG) at the current position\n"
          "  -p, --position [t|b|c][l|r|c]   alignment for later dropons (default tl)\n"
          "  -m, --offset x[,y]        pixel offset for later dropons\n"
          "  -y, --luminance v   -b, --tintblue v   -r, --tintred v   -x, --pixelate   -g, --grayscale\n"
          "  -O, --optimize   -P, --progressive   -A, --arithmetric     options for later --output\n"
          "batch mode:\n"
          "  --batch list.txt --outdir dir [--jobs N] [--devices G]\n"
          "                            run the script (everything but -i/-o) on every listed file with N threads\n"
          "                            spread over G GPUs (default: all)\n",
          stderr);
}

/* one action; returns 0 to go on, a process exit code otherwise (messages as the reference's) */
static int run_action(state_t *s, const action_t *a) {
    char *comma;
    int   rv;

    switch(a->code) {
        case 'i':
            if(mj_read_jpeg_from_file(&s->image, a->arg, 0) != MJ_OK) {
                fprintf(stderr, "Can't read image from '%s'\n", a->arg);
                return 1;
            }
            return 0;
        case 'o':
            if(mj_write_jpeg_to_file(&s->image, a->arg, s->write_options) != MJ_OK) {
                fprintf(stderr, "Can't write image to '%s'\n", a->arg);
                return 1;
            }
            return 0;
        case 'd': {
            char *copy = strdup(a->arg); /* the script is reused per file in batch mode */
            if(copy == NULL) return 1;
            comma = strchr(copy, ',');
            if(comma != NULL) *comma = '\0';
            rv = mj_read_dropon_from_file(&s->dropon, copy, comma != NULL ? comma + 1 : NULL, MJ_BLEND_FULL);
            if(rv != MJ_OK) {
                fprintf(stderr, "Can't read dropon from '%s'\n", copy);
                free(copy);
                return 1;
            }
            free(copy);
            if(mj_compose(&s->image, &s->dropon, s->position, s->offset_x, s->offset_y) != MJ_OK) {
                fprintf(stderr, "Failed to apply the dropon onto the image\n");
                return 1;
            }
            return 0;
        }
        case 'p':
            if(strlen(a->arg) != 2) {
                fprintf(stderr, "Invalid position, use --help for more details\n");
                return 0;
            }
            s->position = 0;
            s->position |= a->arg[0] == 't' ? MJ_ALIGN_TOP : (a->arg[0] == 'b' ? MJ_ALIGN_BOTTOM : (a->arg[0] == 'c' ? MJ_ALIGN_CENTER : 0));
            s->position |= a->arg[1] == 'l' ? MJ_ALIGN_LEFT : (a->arg[1] == 'r' ? MJ_ALIGN_RIGHT : (a->arg[1] == 'c' ? MJ_ALIGN_CENTER : 0));
            return 0;
        case 'm':
            s->offset_x = (int)strtol(a->arg, NULL, 10);
            comma = strchr(a->arg, ',');
            if(comma != NULL) s->offset_y = (int)strtol(comma + 1, NULL, 10); /* no comma: y keeps its value */
            return 0;
        case 'y': (void)mj_effect_luminance(&s->image, (int)strtol(a->arg, NULL, 10)); return 0;
        case 'b': (void)mj_effect_tint(&s->image, (int)strtol(a->arg, NULL, 10), 0); return 0;
        case 'r': (void)mj_effect_tint(&s->image, 0, (int)strtol(a->arg, NULL, 10)); return 0;
        case 'x': (void)mj_effect_pixelate(&s->image); return 0;
        case 'g': (void)mj_effect_grayscale(&s->image); return 0;
        case 'O': s->write_options |= MJ_OPTION_OPTIMIZE; return 0;
        case 'P': s->write_options |= MJ_OPTION_PROGRESSIVE; return 0;
        case 'A': s->write_options |= MJ_OPTION_ARITHMETRIC; return 0;
        default: return 0;
    }
}

/* ---- batch mode ---------------------------------------------------------------------- */
static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

typedef struct {
    char          **files;
    double          t_read, t_script, t_write; /* summed over the workers (MJ_AMD=cli-timing) */
    int             nfiles, next, failed;
    const action_t *script;
    int             nscript;
    const char     *outdir;
    int             ndevices, started; /* worker t works on device t mod ndevices */
    pthread_mutex_t lock;
} batch_t;

/* drop-in internals (shadow.c) */
int mjh_device_count(void);
int mjh_use_device(int ordinal);

static void *batch_worker(void *arg) {
    batch_t *b = (batch_t *)arg;
    int      me;
    pthread_mutex_lock(&b->lock);
    me = b->started++;
    pthread_mutex_unlock(&b->lock);
    if(b->ndevices > 1 && mjh_use_device(me % b->ndevices) != MJ_OK) {
        fprintf(stderr, "modjpeg: worker %d cannot use GPU %d\n", me, me % b->ndevices);
        pthread_mutex_lock(&b->lock);
        b->failed++;
        pthread_mutex_unlock(&b->lock);
        return NULL;
    }
    for(;;) {
        state_t  s;
        action_t io;
        char     out[4096];
        int      k, i, bad = 0;
        double   t0, t1, t2, t3;

        pthread_mutex_lock(&b->lock);
        k = b->next++;
        pthread_mutex_unlock(&b->lock);
        if(k >= b->nfiles) return NULL;

        state_init(&s);
        t0 = now_ms();
        io.code = 'i';
        io.arg = b->files[k];
        bad = run_action(&s, &io);
        t1 = now_ms();
        for(i = 0; i < b->nscript && !bad; i++) {
            if(b->script[i].code != 'i' && b->script[i].code != 'o') bad = run_action(&s, &b->script[i]);
        }
        t2 = now_ms();
        if(!bad) {
            const char *base = strrchr(b->files[k], '/');
            snprintf(out, sizeof(out), "%s/%s", b->outdir, base != NULL ? base + 1 : b->files[k]);
            io.code = 'o';
            io.arg = out;
            bad = run_action(&s, &io);
        }
        state_free(&s);
        t3 = now_ms();
        pthread_mutex_lock(&b->lock);
        if(bad) b->failed++;
        b->t_read += t1 - t0;
        b->t_script += t2 - t1;
        b->t_write += t3 - t2;
        pthread_mutex_unlock(&b->lock);
    }
}

/* drop-in internal (shadow.c): images stay on the device between read and write.  Safe here
 * because this program only ever touches an image through the mj_* functions. */
void mjh_set_device_resident(int on);
int  mjh_option_named(const char *token); /* host/options.c: a token of MJ_AMD */

static int run_batch(const char *list, const char *outdir, int jobs, int devices, const action_t *script, int nscript) {
    batch_t    b;
    FILE      *f = strcmp(list, "-") == 0 ? stdin : fopen(list, "r");
    char       line[4096];
    pthread_t *th;
    int        cap = 0, t;
    struct timespec t0, t1;

    if(f == NULL) {
        fprintf(stderr, "Can't read the batch list '%s'\n", list);
        return 1;
    }
    memset(&b, 0, sizeof(b));
    while(fgets(line, sizeof(line), f) != NULL) {
        size_t n = strlen(line);
        while(n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = '\0';
        if(n == 0) continue;
        if(b.nfiles == cap) {
            cap = cap ? cap * 2 : 64;
            b.files = (char **)realloc(b.files, sizeof(char *) * (size_t)cap);
        }
        b.files[b.nfiles++] = strdup(line);
    }
    if(f != stdin) fclose(f);
    b.script = script;
    b.nscript = nscript;
    b.outdir = outdir;
    pthread_mutex_init(&b.lock, NULL);
    if(jobs < 1) jobs = 1;
    if(jobs > b.nfiles) jobs = b.nfiles > 0 ? b.nfiles : 1;
    {
        const int present = mjh_device_count();
        b.ndevices = (devices > 0 && devices < present) ? devices : present;
        if(b.ndevices > jobs) b.ndevices = jobs;
    }

    clock_gettime(CLOCK_MONOTONIC, &t0);
    th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)jobs);
    for(t = 0; t < jobs; t++) pthread_create(&th[t], NULL, batch_worker, &b);
    for(t = 0; t < jobs; t++) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    {
        const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        fprintf(stderr, "modjpeg: %d files, %d failed, %d threads on %d GPU%s, %.3f s, %.1f images/s\n", b.nfiles, b.failed, jobs, b.ndevices, b.ndevices == 1 ? "" : "s", sec,
                sec > 0 ? b.nfiles / sec : 0.0);
        if(mjh_option_named("cli-timing") && b.nfiles > 0) {
            extern int mjh_shadow_count, mjh_gpu_decode_count, mjh_gpu_encode_count;
            fprintf(stderr, "modjpeg: per file, in the worker that had it: read %.3f ms, script %.3f ms, write+free %.3f ms\n", b.t_read / b.nfiles, b.t_script / b.nfiles,
                    b.t_write / b.nfiles);
            fprintf(stderr, "modjpeg: %d GPU entropy decodes, %d kept on the device, %d GPU entropy encodes\n", mjh_gpu_decode_count, mjh_shadow_count, mjh_gpu_encode_count);
        }
    }
    free(th);
    for(t = 0; t < b.nfiles; t++) free(b.files[t]);
    free(b.files);
    pthread_mutex_destroy(&b.lock);
    return b.failed ? 1 : 0;
}

int main(int argc, char **argv) {
    action_t   *script = (action_t *)calloc((size_t)argc + 1, sizeof(action_t));
    const char *batch_list = NULL, *outdir = ".";
    int         nscript = 0, jobs = 4, devices = 0, c, i, rc = 0;

    /* this program only ever touches an image through the mj_* functions, so images may stay on the
     * device between read and write (MJ_AMD=no-device-resident keeps the host arrays authoritative) */
    mjh_set_device_resident(!mjh_option_named("no-device-resident")); /* MJ_AMD=no-device-resident: images travel through libjpeg's host arrays as in the API */
    opterr = 1;
    /* leading ':' => a missing argument is reported as ':' like the reference's option string */
    while((c = getopt_long(argc, argv, ":i:o:d:p:m:y:b:r:xgPOAh", long_options, NULL)) != -1) {
        switch(c) {
            case 'h': usage(); free(script); return 0;
            case ':': fprintf(stderr, "Argument missing, use --help for more details\n"); break;
            case '?': fprintf(stderr, "Unknown option, use --help for more details\n"); break;
            case OPT_BATCH: batch_list = optarg; break;
            case OPT_OUTDIR: outdir = optarg; break;
            case OPT_JOBS: jobs = atoi(optarg); break;
            case OPT_DEVICES: devices = atoi(optarg); break;
            default:
                script[nscript].code = c;
                script[nscript].arg = optarg;
                nscript++;
                break;
        }
    }

    if(batch_list != NULL) {
        rc = run_batch(batch_list, outdir, jobs, devices, script, nscript);
    }
    else {
        state_t s;
        state_init(&s);
        for(i = 0; i < nscript && rc == 0; i++) rc = run_action(&s, &script[i]);
        state_free(&s);
    }
    free(script);
    return rc;
}
