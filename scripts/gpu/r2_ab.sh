#!/bin/bash
timeout 2000 python scripts/gpu/ab_step.py 2 "t432:" "t256:PPF_SPLITK_TARGET=256" "t320:PPF_SPLITK_TARGET=320" "t384:PPF_SPLITK_TARGET=384"
