// The one place where the library reads the process environment (env.h).
#include "env.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace texir {

static int geti(const char* name, int dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

static Env parse()
{
    Env v{};
    v.bvh_width = geti("TEXIR_BVH_WIDTH", 4);
    v.bvh_layout = std::min(3, std::max(0, geti("TEXIR_BVH_LAYOUT", 0)));
    v.uniform_float = geti("TEXIR_UNIFORM_FLOAT", 1) != 0;
    v.tex_layout = geti("TEXIR_TEX_LAYOUT", 4);
    const int w = geti("TEXIR_SCHED_WEIGHT", 0);
    v.sched_weight = (w >= 1 && w <= 4) ? w : 0;
    v.mip_per_level = geti("TEXIR_MIP_PER_LEVEL", 0) != 0;
    v.adam_scalar = getenv("TEXIR_ADAM_SCALAR") != nullptr;          // (presence switches it on, as before)
    v.adam_grid_y = std::max(0, geti("TEXIR_ADAM_GRID_Y", 0));
    const int ml = geti("TEXIR_MAX_LEAF", 0);
    v.max_leaf = ml ? std::min(8, std::max(1, ml)) : 0;
    v.box_slack_log2 = geti("TEXIR_BOX_SLACK_LOG2", -19);
    const int pw = geti("TEXIR_IRT_TEXELS_PER_WAVE", 0);
    v.irt_texels_per_wave = (pw == 1 || pw == 64) ? pw : 0;
    const int rf = geti("TEXIR_IRT_REFILL", 0);
    v.irt_refill = (rf >= 1 && rf <= 63) ? rf : 0;
    const int mc = geti("TEXIR_IRT_MIN_PART_CELLS", 8);
    v.irt_min_part_cells = mc >= 1 ? mc : 8;
    v.irt_log2parts_cap = getenv("TEXIR_IRT_LOG2PARTS") ? std::max(0, geti("TEXIR_IRT_LOG2PARTS", 0)) : -1;
    v.irt_grid_cap = std::max(0, geti("TEXIR_IRT_GRID_CAP", 0));
    const int gc = geti("TEXIR_SPEC_GRID_CAP", 1 << 16);
    v.spec_grid_cap = gc >= 1 ? gc : (1 << 16);
    const int lpp = geti("TEXIR_SPEC_LPP", 0);
    v.spec_lpp = (lpp >= 1 && lpp <= 64 && (lpp & (lpp - 1)) == 0) ? lpp : 0;
    return v;
}

static Env g_env = parse();          // library load

const Env& env() { return g_env; }
void env_reload() { g_env = parse(); }

int env_switch(const char* name, int* value)
{
    const struct { const char* n; int v; } tab[] = {
        {"TEXIR_BVH_WIDTH", g_env.bvh_width}, {"TEXIR_BVH_LAYOUT", g_env.bvh_layout}, {"TEXIR_UNIFORM_FLOAT", g_env.uniform_float},
        {"TEXIR_TEX_LAYOUT", g_env.tex_layout}, {"TEXIR_SCHED_WEIGHT", g_env.sched_weight}, {"TEXIR_MIP_PER_LEVEL", g_env.mip_per_level},
        {"TEXIR_ADAM_SCALAR", g_env.adam_scalar}, {"TEXIR_ADAM_GRID_Y", g_env.adam_grid_y}, {"TEXIR_MAX_LEAF", g_env.max_leaf},
        {"TEXIR_BOX_SLACK_LOG2", g_env.box_slack_log2}, {"TEXIR_IRT_TEXELS_PER_WAVE", g_env.irt_texels_per_wave}, {"TEXIR_IRT_REFILL", g_env.irt_refill},
        {"TEXIR_IRT_MIN_PART_CELLS", g_env.irt_min_part_cells}, {"TEXIR_IRT_LOG2PARTS", g_env.irt_log2parts_cap},
        {"TEXIR_IRT_GRID_CAP", g_env.irt_grid_cap}, {"TEXIR_SPEC_GRID_CAP", g_env.spec_grid_cap}, {"TEXIR_SPEC_LPP", g_env.spec_lpp}};
    for (const auto& t : tab)
        if (name && !strcmp(name, t.n)) { *value = t.v; return 0; }
    return -1;
}

}  // namespace texir
