"""Hot-path options.

Mirrors the globals of the reference's common.cpp:79-180 that change the results
of segment_juncs / long_spanning_reads.  Field order is
the field order of `thj_params` in include/thj.h (and `orc_params` in
oracle/thj_oracle.h for the first twelve).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, fields

LIBRARY_TYPES = {  # common.h:155-167
    "": 0, "none": 0,
    "fr-unstranded": 1, "fr-firststrand": 2, "fr-secondstrand": 3,
    "ff-unstranded": 4, "ff-firststrand": 5, "ff-secondstrand": 6,
}

READ_LEFT, READ_RIGHT = 1, 2  # segments.h:13-18


@dataclass
class Params:
    segment_length: int = 25          # common.cpp:121
    segment_mismatches: int = 2       # common.cpp:122
    min_segment_intron: int = 50      # common.cpp:115
    max_segment_intron: int = 500000  # common.cpp:116
    max_insertion_length: int = 3     # common.cpp:98
    max_deletion_length: int = 3      # common.cpp:99
    max_seg_multihits: int = 40       # common.cpp:135
    inner_dist_mean: int = 200        # common.cpp:101
    inner_dist_std_dev: int = 20      # common.cpp:102
    library_type: int = 0             # common.cpp:180
    bowtie2: int = 1                  # common.cpp:79
    read_side: int = READ_LEFT
    # long_spanning_reads
    min_report_intron: int = 50       # common.cpp:106
    max_report_intron: int = 500000   # common.cpp:107
    min_anchor_len: int = 8           # common.cpp:105
    read_mismatches: int = 2          # common.cpp:123
    read_gap_length: int = 2          # common.cpp:124
    read_edit_dist: int = 2           # common.cpp:125
    bowtie2_max_penalty: int = 6      # common.cpp:87
    bowtie2_min_penalty: int = 2      # common.cpp:88
    bowtie2_penalty_for_N: int = 1    # common.cpp:89
    bowtie2_read_gap_open: int = 5    # common.cpp:90
    bowtie2_read_gap_cont: int = 3    # common.cpp:91
    bowtie2_ref_gap_open: int = 5     # common.cpp:92
    bowtie2_ref_gap_cont: int = 3     # common.cpp:93
    fusion_anchor_length: int = 20    # common.cpp:172
    fusion_min_dist: int = 10000000   # common.cpp:173
    fusion_search: int = 0            # common.cpp:171

    def as_ctypes(self) -> "CParams":
        c = CParams()
        for f in fields(self):
            setattr(c, f.name, int(getattr(self, f.name)))
        return c


class CParams(ctypes.Structure):
    _fields_ = [(f.name, ctypes.c_int32) for f in fields(Params)]
