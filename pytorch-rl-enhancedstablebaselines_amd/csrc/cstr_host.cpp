// cstr_host.cpp -- host-only pieces of the C ABI (no device code).
#include <hip/hip_runtime_api.h>

#include "../../include/cstr_rl_hip.h"

extern "C" int cstr_abi_version(void) { return CSTR_ABI_VERSION; }

extern "C" const char *cstr_error_string(int code)
{
    switch (code) {
    case CSTR_OK: return "ok";
    case CSTR_E_BADARG: return "cstr: bad argument (null pointer, non-positive size or misaligned buffer)";
    case CSTR_E_UNSUPPORTED: return "cstr: unsupported configuration (obs_dim must be 4 or 8, act_dim 2, batch <= 16384, 32-bit index range)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "cstr: unknown error";
    }
}

// Class constants of TwoSeriesCSTREnv (twoseriescstr.py:37-61) folded the way NumPy >= 2 evaluates
// _dynamics: Python-float sub-expressions in double first, then rounded to f32 where they meet an f32.
extern "C" void cstr_default_coef(cstr_coef_t *c, double target_c2, double min_conc, double max_conc, int32_t max_steps)
{
    const double Q = 50, V1 = 100, V2 = 100, Cf = 0.5, Tf = 320, Tcf = 370, k0 = 7.2e10, E = 8.314e4, R = 8.314;
    const double dH = -6.78e4, rou = 1000, rou_c = 1000, c_p = 0.239, c_pc = 0.239, U = 6.6e5, A1 = 8.958, A2 = 8.958;
    c->q_v1 = (float)(Q / V1);
    c->q_v2 = (float)(Q / V2);
    c->cf = (float)Cf;
    c->tf = (float)Tf;
    c->tcf = (float)Tcf;
    c->k0 = (float)k0;
    c->neg_e = (float)(-E);
    c->r_gas = (float)R;
    c->hk = (float)(-dH * k0);
    c->rho_cp = (float)(rou * c_p);
    c->cool1 = (float)((rou_c * c_pc) / (rou * c_p * V1));
    c->cool2 = (float)((rou_c * c_pc) / (rou * c_p * V2));
    c->neg_ua1 = (float)(-(U * A1));
    c->neg_ua2 = (float)(-(U * A2));
    c->rho_c = (float)rou_c;
    c->c_pc = (float)c_pc;
    c->dt = (float)0.1;
    const float lo[4] = {0.0f, 273.15f, 0.0f, 273.15f}, hi[4] = {0.7f, 400.0f, 0.7f, 400.0f};
    for (int i = 0; i < 4; ++i) {
        c->s_lo[i] = lo[i];
        c->s_hi[i] = hi[i];
        c->s_span[i] = hi[i] - lo[i];
    }
    for (int i = 0; i < 2; ++i) {
        c->a_lo[i] = 30.0f;
        c->a_hi[i] = 250.0f;
        c->a_span[i] = 250.0f - 30.0f;
    }
    c->target_c2 = (float)target_c2;
    c->conc_span = (float)(max_conc - min_conc);
    c->max_steps = max_steps;
}
