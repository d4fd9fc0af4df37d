// Host-side TPWL handle (owns the HBM tables) shared by tpwl.hip / lqr.hip / scp.hip.
#pragma once
#include "common.h"
#include "tpwl_dev.h"

struct stpwl {
    int P = 0, r = 0, n = 0, m = 0, nz = 0;
    double w_q = 1.0, w_v = 0.0;
    bool has_discrete = false;
    srh::DevBuf qT, vT, u, Ac, Bc, dc, AcT, BcT, Ad, Bd, dd, AdT, BdT, H, z_ref;
    std::vector<double> H_host, zref_host;
    TpwlDev view() const;
};
