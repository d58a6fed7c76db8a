#!/bin/bash
# Per-stage count of register-file shuffles and hazard nops in a listing of eval_t16.hip (MODE 0 kernel):
#   tools/t16_shuffles.sh file.s
awk '/^_ZN2bf3t1612_GLOBAL__N_110k_eval_t16ILi0E.*:/{f=1} f&&/v_mfma_f32_16x16x32/{m++} f{ if(m<284) st="1 stage1"; else if(m<1820) st="2 pair"; else if (m<2468) st="3 conv"; else st="4 head"; if ($1 ~ /v_accvgpr_write/) w[st]++; if ($1 ~ /v_accvgpr_read/) r[st]++; if ($1 ~ /v_accvgpr_mov/) mv[st]++; if ($1=="s_nop") {n[st]++; nn[st]+=$2+1}; if ($1 ~ /^scratch_/) sc[st]++; tot[st]++} f&&/s_endpgm/{exit} END{for (s in tot) print s, "instrs", tot[s], "acc_write", w[s]+0, "acc_read", r[s]+0, "acc_mov", mv[s]+0, "s_nop", n[s]+0, "nop cycles", nn[s]+0, "scratch", sc[s]+0}' "$1" | sort
grep -A95 "k_eval_t16ILi0E" "$1" | grep -m2 "vgpr_spill_count\|private_segment_fixed_size:"
