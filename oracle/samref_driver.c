/* samref_driver.c -- TEST INFRASTRUCTURE, build container only (never shipped, never linked into the product).
 *
 * A thin command-line driver over the reference's own vendored samtools 0.1.18 (src/samtools-0.1.18: plain C + zlib), which
 * oracle/ref_samtools.mk compiles from the sources where they lie under /root/reference into oracle/_ref/.  It does what the
 * reference's regression harness does with its outputs (tests/regression_tests/regression_test.py:78-118: `samtools view`,
 * `samtools calmd`) plus the SAM -> BAM import, so that tests/golden/ref_samtools/ can hold vectors made by REFERENCE code:
 *
 *   samref view   in.bam            records as bam_format1() prints them (sam.c / bam.c)
 *   samref sam2bam in.sam out.bam   sam_read1 -> bam_write1 through bgzf.c at zlib's default level: the BAM byte stream and BGZF
 *                                    members that SURVEY 8(a) row B8 and the inflate kernels are checked against
 *   samref calmd  in.bam ref.fa     MD / NM of every record recomputed by bam_fillmd1_core (bam_md.c:23-131), printed beside the
 *                                    record's own
 * Only this file is ours; everything it calls is the reference's. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sam.h"
#include "faidx.h"

void bam_fillmd1_core(bam1_t *b, char *ref, int flag, int max_nm);
#define UPDATE_NM 8
#define UPDATE_MD 16

static int cmd_view(const char *in)
{
    samfile_t *fp = samopen(in, "rb", 0);
    if (!fp) { fprintf(stderr, "samref: cannot open %s\n", in); return 1; }
    bam1_t *b = bam_init1();
    while (samread(fp, b) >= 0) { char *s = bam_format1(fp->header, b); puts(s); free(s); }
    bam_destroy1(b); samclose(fp);
    return 0;
}

static int cmd_sam2bam(const char *in, const char *out)
{
    samfile_t *fi = samopen(in, "r", 0);
    if (!fi || !fi->header) { fprintf(stderr, "samref: cannot open %s (a SAM file with @SQ lines)\n", in); return 1; }
    samfile_t *fo = samopen(out, "wb", fi->header);
    if (!fo) { fprintf(stderr, "samref: cannot write %s\n", out); return 1; }
    bam1_t *b = bam_init1();
    while (samread(fi, b) >= 0) samwrite(fo, b);
    bam_destroy1(b); samclose(fo); samclose(fi);
    return 0;
}

static int cmd_calmd(const char *in, const char *fa)
{
    samfile_t *fp = samopen(in, "rb", 0);
    faidx_t *fai = fai_load(fa);
    if (!fp || !fai) { fprintf(stderr, "samref: cannot open %s / %s\n", in, fa); return 1; }
    bam1_t *b = bam_init1();
    char *ref = 0; int tid = -1, len = 0;
    while (samread(fp, b) >= 0) {
        if (b->core.tid < 0) continue;
        if (b->core.tid != tid) { free(ref); ref = fai_fetch(fai, fp->header->target_name[b->core.tid], &len); tid = b->core.tid; }
        uint8_t *p;
        char old_md[4096] = "-"; long old_nm = -1;
        if ((p = bam_aux_get(b, "MD")) != 0) { snprintf(old_md, sizeof old_md, "%s", (char*)p + 1); bam_aux_del(b, p); }
        if ((p = bam_aux_get(b, "NM")) != 0) { old_nm = bam_aux2i(p); bam_aux_del(b, p); }
        if (ref) bam_fillmd1_core(b, ref, UPDATE_NM | UPDATE_MD, 0);
        const char *md = (p = bam_aux_get(b, "MD")) ? (char*)p + 1 : "-";
        long nm = (p = bam_aux_get(b, "NM")) ? bam_aux2i(p) : -1;
        printf("%s\t%d\t%s\t%d\t%s\t%ld\t%s\t%ld\n", bam1_qname(b), b->core.flag, fp->header->target_name[b->core.tid], b->core.pos + 1, md, nm, old_md, old_nm);
    }
    free(ref); bam_destroy1(b); fai_destroy(fai); samclose(fp);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 3 && !strcmp(argv[1], "view")) return cmd_view(argv[2]);
    if (argc == 4 && !strcmp(argv[1], "sam2bam")) return cmd_sam2bam(argv[2], argv[3]);
    if (argc == 4 && !strcmp(argv[1], "calmd")) return cmd_calmd(argv[2], argv[3]);
    fprintf(stderr, "usage: samref view in.bam | sam2bam in.sam out.bam | calmd in.bam ref.fa\n");
    return 2;
}
