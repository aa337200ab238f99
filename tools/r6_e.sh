#!/bin/bash
# round 6, fifth lease: forms on top of the static body (27 = static baseline as a form; 28 / 31 pieces priced 4 / 12; 29 / 30 pre 0 / 48)
FORMS="0 27 28 31 29 30" ROUNDS=6 bash tools/r6_forms.sh r6f4
