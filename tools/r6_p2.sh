#!/bin/bash
tools/plus_timeline.sh r6_tl_default
