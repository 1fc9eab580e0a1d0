#!/bin/bash
# VGPRs / scratch / occupancy of every kernel instantiation (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/.."
make -C software-raytracer_amd/csrc -s asm 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | sed -E 's/.*(Function Name: |VGPRs: |SGPRs: |ScratchSize \[bytes\/lane\]: |Occupancy \[waves\/SIMD\]: |LDS Size \[bytes\/block\]: )/\1/; s/ \[-Rpass.*//' | paste - - - - - - | sed -E 's/_ZN3srt//; s/EEvNS_12KernelParamsE.*Function/ Function/'
