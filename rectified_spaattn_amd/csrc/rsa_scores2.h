// K2, second form (rsa_scores2.hip): arguments and the launch hook used by rsa_stats.hip::rsa_pooled_scores.
#pragma once
#include "rsa_common.h"

struct Score2Args {
    const float *qbar, *aq, *kbar, *ak;   // [BH, NBv, D] fp32 (K1)
    const unsigned short* ktxt;           // K base pointer (text-token columns)
    long ksb, ksh, kss;
    float* scores;                        // [BH, NBv, NS]
    uint8_t* unrel;                       // [BH, NBv, NBv]
    int NBv, n_txt, NS, H, BH;
    // filled by the launcher: i tiles of 32 rows, visual / all j sub-tiles of 32 columns, j ranges per i tile, items
    int nti, ntj0, ntj, JS, n_items;
    int form;                             // 0 = the product; other values: timing-only forms (tuning key "k2_form")
};

int rsa_launch_pooled_scores2(Score2Args a, int D, int dtype, hipStream_t s);
