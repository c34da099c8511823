#!/bin/bash
timeout 300 python tools/diag_split_attn.py 2>&1 | tail -22
