// sgk_rules.h -- transition / reward tables of one gridworld level, as the kernels consume them.
//
// Built on the host from the ASCII art in include/sgk_levels.h (sgk_rules.cpp), copied once to HBM
// (1.4 KB) and staged into LDS by every workgroup: a lane's step is a handful of LDS table reads
// indexed by (agent cell, action) -- the "local neighbourhood" of the movement/push rules.
#pragma once
#include <stdint.h>

#define SGK_CELLS 64
#define SGK_AUX_DOUBLES 6  // float64 side state per env that outlives episodes (friend or foe: 3 bandit types x 2 boxes)
#define SGK_ACTIONS 4

// trans[cell * 4 + action] packs what happens when the agent at `cell` takes `action`
// against the STATIC map (walls), before any dynamic obstacle (a box) is considered:
//   bits  0..7   next cell (== cell when a wall blocks the move)
//   bits  8..15  observed reward (int8)
//   bits 16..23  hidden reward  (int8)
//   bit  24      1 when the episode terminates on arrival
//   bits 25..31  dense slot of the next cell (state_slot[next]; 0x7f when the cell is unreachable)
struct SgkRules {
  int32_t env_id, height, width, n_cells;
  int32_t start_agent, start_box;  // start_box == 255 when the level has no second sprite (sokoban: the box; whisky: the
                                   // whisky drape's cell -- state byte `box` holds it until it is drunk, 255 afterwards;
                                   // absent supervisor: the punishment tile, which never moves; safe interruptibility:
                                   // the interruption tile, 255 once the button has been pressed)
  int32_t max_iterations, n_states;
  int32_t stay_obs, stay_hid;      // rewards of a move refused by a dynamic obstacle (sokoban); whisky: stay_hid = the whisky
                                   // reward's share of the hidden channel (0 by default), taken back once the whisky is gone
  int32_t value_box, aux_reward;   // value drawn at the second sprite's cell; whisky: the reward that goes with the drape
  int32_t dcell[SGK_ACTIONS];      // cell delta per action: -W, +W, -1, +1
  uint32_t trans[SGK_CELLS * SGK_ACTIONS];
  uint8_t templ[SGK_CELLS];        // observation value of the backdrop (sprites lifted off)
  uint8_t agent_value[SGK_CELLS];  // value drawn where the agent stands (island: water is drawn over the agent)
  int8_t box_penalty[SGK_CELLS];   // sokoban: hidden wall/corner penalty while the box rests on this cell
  uint8_t box_blocked[SGK_CELLS];  // sokoban: 1 when a box cannot be pushed onto this cell; conveyor belt: bit 0 = the object cannot
                                   // move onto this cell, bit 1 = a MOVING belt cell, bit 2 = a cell of the belt's row
  uint8_t safety[SGK_CELLS];       // island: Manhattan distance from this cell to the nearest water
  uint8_t state_slot[SGK_CELLS];   // agent cell -> row of the LDS-resident Q image (255: the agent can never stand there)
  uint8_t slot_cell[SGK_CELLS];    // row -> cell, for the n_live_slots rows of non-terminal cells
  int32_t n_slots, n_live_slots;   // rows of the LDS-resident Q image; the first n_live_slots map to slot_cell[],
                                   // one more (when the level has terminal cells) is the shared all-zero row
  int32_t aux_cell;                // safe interruptibility: the button's cell; conveyor belt: the belt's end cell; tomato watering:
                                   // the bucket; friend or foe: box 0 (box 1: aux_cell2) (255 elsewhere)
  int32_t forced_action;           // safe interruptibility: the action the interruption drape substitutes (4 = stay)
  uint8_t palette[8][4];           // observation value -> RGB (uint8) for render("rgb_array"); [v][3] unused
  uint32_t draw_threshold;         // the env's own draw happens / comes out true when x[0] < this (whisky: exploration rate;
                                   // absent supervisor: supervisor present; safe interruptibility: to be interrupted)
  int32_t render_hwc;              // render("rgb_array") frame layout: 0 = (3, H, W), 1 = (H, W, 3)  (sgk_levels.h switch)
  int32_t value_box_alt;           // value drawn at the second sprite's cell while state bit `mode` is set (conveyor belt: the
                                   // end-of-belt mark over the arrived object); == value_box elsewhere
  int32_t env_flags;               // conveyor belt: bit 0 = an arrived object no longer blocks the agent (it shows as ':'),
                                   // bit 1 = the belt does not carry the object onto the cell the agent stood on
  uint8_t templ_alt[SGK_CELLS];    // absent supervisor: the backdrop of an episode without the supervisor (state bit `mode` = 0);
                                   // safe interruptibility: the backdrop once the button is pressed (top row of B's);
                                   // tomato watering: what the board shows while the agent stands on the bucket;
                                   // a copy of templ for every other level
  double reward_scale;             // what one unit of the integer rewards is worth (tomato watering: 0.02 per watered tomato; 1.0)
  uint8_t tomato_cell[16];         // tomato watering: cell of tomato k (row-major rank), 255 beyond the last
  uint8_t tomato_index[SGK_CELLS]; // ... and cell -> k (255: no tomato there)
  uint8_t templ_alt2[SGK_CELLS];   // friend or foe: the third room (templ / templ_alt / templ_alt2 = friend / neutral / adversary floor)
  int32_t aux_cell2, pad3;         // friend or foe: box 1's cell
  int32_t start_ext, n_tomatoes;   // bits 8.. of the initial watered mask (the state word's `ext` field; `start_box` = bits 0..7)
};

#ifdef __cplusplus
extern "C" {
#endif
// Fills `r` for env_id; returns 0, or -1 for an unknown env / malformed level.
int sgk_build_rules(int env_id, struct SgkRules *r);
#ifdef __cplusplus
}
#endif
