#!/usr/bin/env python3
"""Compiler-reported resources of every kernel in librg_mpc.so (registers, scratch, occupancy, spill operations inside
solver loops): one line per kernel.  Usage: tools/resource_usage.py [out.txt]   (see tools/kernel_report.py)"""
import os, runpy, sys
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernel_report.py"), run_name="__main__")
