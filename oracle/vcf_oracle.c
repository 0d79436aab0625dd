/*
 * oracle/vcf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * calculate_score (PileupModel/predict.py:31-34, HaplotypeModel/predict_dev.py:21-24):
 *   tmp = max((-10 * log(e, 10)) * log(((1.0 - p) + 1e-300) / (p + 1e-300)) + 10, 0); round(tmp, 2)
 * Python's log(e, 10) is log(e)/log(10) in float64; round(x, 2) is correctly-rounded decimal
 * rounding of the double, which printf("%.2f") + strtod reproduces.
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

double orc_calculate_score(double p)
{
    const double log_e_10 = log(M_E) / log(10.0);
    double tmp = (-10.0 * log_e_10) * log(((1.0 - p) + 1e-300) / (p + 1e-300)) + 10.0;
    if (!(tmp > 0.0)) tmp = 0.0;
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", tmp);
    return strtod(buf, NULL);
}
