// Host-side SSM handle (owns the coefficient tables in HBM) shared by ssm.hip / lqr.hip.
#pragma once
#include "common.h"
#include "ssm_dev.h"

struct sssm {
    int n = 0, m = 0, no = 0, nr = 0, ns = 0;
    bool has_discrete = false;
    size_t lds = 0;
    srh::DevBuf er, es, R, Bc, Rd, Bd, Wc, Vc, z_ref, H;
    srh::DevBuf tr[4], ts[4];          // evaluation tables of the two bases (parent, variable, derivative index, levels)
    int order_r = 0, order_s = 0;
    SsmDev view() const;
};

std::vector<int> ssm_exponents(int dim, int order);
