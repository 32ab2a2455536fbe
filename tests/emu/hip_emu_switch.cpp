// TEST INFRASTRUCTURE ONLY: context switch for tests/emu/hip_emu.h fibres (x86-64 SysV).
asm(R"(
.text
.globl lm_emu_switch
.type lm_emu_switch,@function
lm_emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size lm_emu_switch, .-lm_emu_switch
.section .note.GNU-stack,"",@progbits
)");
