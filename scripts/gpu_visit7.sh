#!/bin/bash
python scripts/exp_bitwise_more.py
