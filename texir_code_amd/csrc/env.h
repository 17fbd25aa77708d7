// Every TEXIR_* run-time switch of the library, parsed ONCE (library load) into one struct: no launch path calls getenv.
// texir_reload_env() (C-ABI, include/texir_hip.h) re-parses the environment -- the test suite flips switches between launches.
#pragma once

namespace texir {

struct Env {
    int bvh_width;             // TEXIR_BVH_WIDTH            4 (default) | 2 = binary tree only
    int bvh_layout;            // TEXIR_BVH_LAYOUT           0 depth-first | 1 sibling blocks | 2, 3 treelets (bvh_build.cpp relayout4)
    int uniform_float;         // TEXIR_UNIFORM_FLOAT        1 (default) | 0 = scenes without the float node copy (every node step per lane)
    int tex_layout;            // TEXIR_TEX_LAYOUT           4 (default) | 3: 4-byte texels when the texture packs exactly, else 2; 2 | 1 | 0 force the float32 layouts
    int sched_weight;          // TEXIR_SCHED_WEIGHT         0 (default: measured per scene by texir_scene_tune) | 1 | 2 forced
    int mip_per_level;         // TEXIR_MIP_PER_LEVEL        0 | 1 = one launch per mip level (reference form kept for the parity tests)
    int adam_scalar;           // TEXIR_ADAM_SCALAR          0 | 1 = scalar Adam kernel (reference form kept for the parity tests)
    int adam_grid_y;           // TEXIR_ADAM_GRID_Y          0 = full grid | rows of blocks (probe)
    int max_leaf;              // TEXIR_MAX_LEAF             0 = builder default | 1..8 triangles per leaf (quad leaves, the default build: capped at 4)
    int box_slack_log2;        // TEXIR_BOX_SLACK_LOG2       -19 (default) | 99 = no slack
    int irt_texels_per_wave;   // TEXIR_IRT_TEXELS_PER_WAVE  0 = automatic | 1 | 64
    int irt_refill;            // TEXIR_IRT_REFILL           0 = lock-step passes (default) | 1..63: refill idle lanes once this many have gathered (irt_stream_kernel)
    int irt_min_part_cells;    // TEXIR_IRT_MIN_PART_CELLS   8 (default)
    int irt_log2parts_cap;     // TEXIR_IRT_LOG2PARTS        -1 = no cap
    int irt_grid_cap;          // TEXIR_IRT_GRID_CAP         0 = every co-resident workgroup (default) | blocks of the persistent IrT grid (256 = one wave per SIMD: occupancy sweeps, tools/chain_probe.py)
    int spec_grid_cap;         // TEXIR_SPEC_GRID_CAP        65536 (default)
    int spec_lpp;              // TEXIR_SPEC_LPP             0 = automatic | forced lanes per pixel (power of two)
};

const Env& env();          // the current snapshot
int env_switch(const char* name, int* value);   // the snapshot's value of one switch by its variable name; 0 = known
void env_reload();         // re-read the process environment (not thread-safe against concurrent launches: test use only)

}  // namespace texir
