/* pipe_standin.c -- compiled stand-ins for `ffmpeg` and `ffprobe` on the rawvideo-pipe path (there is no ffmpeg in the
 * image).  TEST / BENCH INFRASTRUCTURE, not product code: bench.py's `c5_pipes` leg and tests/ build it with gcc into a
 * scratch directory as two executables named ffmpeg and ffprobe and put that directory in front of PATH, so that
 * dither_pie_amd.video_processor.VideoProcessor.process_video_streaming (the drop-in of the reference's
 * video_processor.py:172-390) runs unchanged against them.  They stream from / to memory: no file I/O, no codec --
 * what is measured is the pipe plumbing either side of the GPU, not ffmpeg.
 *
 *   ffprobe ... -show_entries <what> ...   answers the four queries get_video_info / _probe_rotation make
 *   ffmpeg  ... -s WxH pipe:1              decoder: DP_STANDIN_FRAMES frames of W x H rgb24 to stdout
 *   ffmpeg  ... -s WxH ... pipe:0 ... OUT  encoder: reads stdin to the end, writes a summary (+ the first frames) to OUT
 *
 * Frame f, byte i of the decoder's stream:  ((uint32)(i + (f % DISTINCT) * frame_bytes) * 2654435761u) >> 24, with the
 * first four bytes of every frame replaced by f (little endian) -- numpy restates it in one line (bench.py).
 * Encoder summary (text, one line, then raw bytes):
 *   "<W>x<H> bytes=<n> frames=<n / frame_bytes> wsum=<sum over frames of (f + 1) * bytesum(frame f) mod 2^64> keep=<k>\n"
 *   followed by the first k frames as received (DP_STANDIN_KEEP, default 2).
 * Environment: DP_STANDIN_W, DP_STANDIN_H (ffprobe's answer), DP_STANDIN_FRAMES, DP_STANDIN_DISTINCT (default 8),
 * DP_STANDIN_KEEP, DP_STANDIN_FPS (default "25/1"). */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static long env_long(const char *name, long dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atol(v) : dflt;
}

static int has_arg(int argc, char **argv, const char *what)
{
    for (int i = 1; i < argc; ++i)
        if (strcmp(argv[i], what) == 0) return i;
    return 0;
}

static void widen(int fd)
{
#ifdef F_SETPIPE_SZ
    (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20); /* best effort: not a pipe (/dev/null, a file) or above pipe-max-size */
#endif
}

static int write_all(int fd, const uint8_t *p, size_t n)
{
    while (n) {
        ssize_t k = write(fd, p, n);
        if (k < 0) {
            if (errno == EINTR) continue;
            return -1;
        }
        p += k;
        n -= (size_t)k;
    }
    return 0;
}

static int probe(int argc, char **argv)
{
    const int at = has_arg(argc, argv, "-show_entries");
    const char *e = (at && at + 1 < argc) ? argv[at + 1] : "";
    const char *fps = getenv("DP_STANDIN_FPS");
    if (strcmp(e, "stream=r_frame_rate") == 0) printf("%s\n", (fps && *fps) ? fps : "25/1");
    else if (strcmp(e, "stream=width,height") == 0) printf("%ld\n%ld\n", env_long("DP_STANDIN_W", 1920), env_long("DP_STANDIN_H", 1080));
    else if (strcmp(e, "stream=duration,nb_frames") == 0) printf("N/A\n%ld\n", env_long("DP_STANDIN_FRAMES", 0));
    else printf("\n"); /* rotation: none */
    return 0;
}

static int parse_size(int argc, char **argv, long *w, long *h)
{
    const int at = has_arg(argc, argv, "-s");
    if (!at || at + 1 >= argc) return -1;
    return sscanf(argv[at + 1], "%ldx%ld", w, h) == 2 ? 0 : -1;
}

static int decoder(int argc, char **argv)
{
    long w, h;
    if (parse_size(argc, argv, &w, &h)) return 2;
    const long frames = env_long("DP_STANDIN_FRAMES", 0), distinct = env_long("DP_STANDIN_DISTINCT", 8);
    const size_t fb = (size_t)w * (size_t)h * 3u;
    const long k = distinct < 1 ? 1 : (distinct > frames && frames > 0 ? frames : distinct);
    uint8_t *buf = (uint8_t *)malloc(fb * (size_t)k);
    if (!buf) return 3;
    for (size_t i = 0; i < fb * (size_t)k; ++i) buf[i] = (uint8_t)(((uint32_t)i * 2654435761u) >> 24);
    widen(1);
    for (long f = 0; f < frames; ++f) {
        uint8_t *p = buf + (size_t)(f % k) * fb;
        uint8_t save[4];
        memcpy(save, p, 4);
        const uint32_t tag = (uint32_t)f;
        memcpy(p, &tag, 4);
        const int rc = write_all(1, p, fb);
        memcpy(p, save, 4);
        if (rc) return 4; /* the reader went away */
    }
    free(buf);
    return 0;
}

static int encoder(int argc, char **argv)
{
    long w, h;
    if (parse_size(argc, argv, &w, &h)) return 2;
    const char *out_path = argv[argc - 1];
    const size_t fb = (size_t)w * (size_t)h * 3u;
    const long keep = env_long("DP_STANDIN_KEEP", 2);
    uint8_t *kept = (uint8_t *)malloc(fb * (size_t)(keep > 0 ? keep : 1));
    const size_t chunk = 1u << 20;
    uint8_t *buf = (uint8_t *)malloc(chunk);
    if (!kept || !buf) return 3;
    widen(0);
    uint64_t total = 0, wsum = 0, fsum = 0;
    size_t in_frame = 0; /* bytes of the current frame seen so far */
    uint64_t frame = 0;
    for (;;) {
        ssize_t n = read(0, buf, chunk);
        if (n < 0) {
            if (errno == EINTR) continue;
            return 4;
        }
        if (n == 0) break;
        size_t off = 0;
        while (off < (size_t)n) {
            size_t take = (size_t)n - off;
            if (take > fb - in_frame) take = fb - in_frame;
            if ((long)frame < keep) memcpy(kept + frame * fb + in_frame, buf + off, take);
            uint64_t s = 0;
            for (size_t i = 0; i < take; ++i) s += buf[off + i];
            fsum += s;
            in_frame += take;
            off += take;
            if (in_frame == fb) {
                wsum += (frame + 1) * fsum;
                fsum = 0;
                in_frame = 0;
                ++frame;
            }
        }
        total += (uint64_t)n;
    }
    FILE *f = fopen(out_path, "wb");
    if (!f) return 5;
    const uint64_t k = frame < (uint64_t)keep ? frame : (uint64_t)(keep > 0 ? keep : 0);
    fprintf(f, "%ldx%ld bytes=%llu frames=%llu wsum=%llu keep=%llu\n", w, h, (unsigned long long)total, (unsigned long long)frame,
            (unsigned long long)wsum, (unsigned long long)k);
    fwrite(kept, 1, (size_t)k * fb, f);
    fclose(f);
    return in_frame == 0 ? 0 : 6; /* a partial frame at the end of the stream */
}

int main(int argc, char **argv)
{
    const char *base = strrchr(argv[0], '/');
    base = base ? base + 1 : argv[0];
    if (strcmp(base, "ffprobe") == 0) return probe(argc, argv);
    if (has_arg(argc, argv, "pipe:1")) return decoder(argc, argv);
    if (has_arg(argc, argv, "pipe:0")) return encoder(argc, argv);
    return 2;
}
