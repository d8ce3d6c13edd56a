# TEST INFRASTRUCTURE, build container only.  Own recipe (not the reference's Makefile): compiles the reference's vendored samtools
# 0.1.18 -- plain C that needs nothing but zlib -- from the sources WHERE THEY LIE under /root/reference into oracle/_ref/ (git-ignored),
# together with oracle/samref_driver.c (ours).  No stand-in headers, no generated code, nothing of the reference is copied into the
# repository.  `make -f oracle/ref_samtools.mk` is a no-op when /root/reference is absent (the GPU box): nothing under tests/, bench.py
# or smoke() needs the tool -- they read the fixtures tests/golden/ref_samtools/mint.py made with it.
REF ?= /root/reference/src/samtools-0.1.18
OUT := $(dir $(lastword $(MAKEFILE_LIST)))_ref
SRC := bgzf kstring bam_aux bam bam_import sam sam_header razf faidx knetfile bam_md kprobaln bam_pileup bam_index
CFLAGS := -O2 -w -D_FILE_OFFSET_BITS=64 -D_LARGEFILE64_SOURCE -D_USE_KNETFILE

ifeq ($(wildcard $(REF)/bam.c),)
all:
	@echo "oracle/ref_samtools.mk: $(REF) not present, nothing to build"
else
all: $(OUT)/samref
$(OUT)/%.o: $(REF)/%.c
	@mkdir -p $(OUT)
	gcc $(CFLAGS) -I$(REF) -c $< -o $@
$(OUT)/samref: $(addprefix $(OUT)/,$(addsuffix .o,$(SRC))) $(OUT)/../samref_driver.c
	gcc $(CFLAGS) -I$(REF) $(OUT)/../samref_driver.c $(addprefix $(OUT)/,$(addsuffix .o,$(SRC))) -o $@ -lz -lm
endif
