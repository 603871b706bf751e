// tests/refdrop: threadIdx / blockIdx come with hip_runtime.h
#pragma once
