// The tier-A/C kernel source, carried inside libdnastore_amd.so for hiprtc (jit.cpp).
    .section .rodata
    .global dnas_tiera_source
    .global dnas_tiera_source_end
dnas_tiera_source:
    .incbin "viterbi_tiera.hip"
dnas_tiera_source_end:
    .byte 0
    .section .note.GNU-stack,"",@progbits
