// HOST code (no kernels): slice_data() of one H.264 picture -- the native form of sleap_amd/io/_h264.py (`_SliceDecoder`,
// `_CavlcSliceDecoder`, `deblock_inter`) and of the intra module it builds on (io/_h264_intra.py), function by function. The
// reference reads video through cv2 / FFmpeg (sleap/io/video.py:340-504: native code); neither exists in this image, the Python
// decoder is the checker (1-5 pictures / s), this file is what `MediaVideo` runs (several hundred). Python keeps what is cheap
// and stateful: MP4 tables, parameter sets, the slice header, picture order counts, reference marking and list construction; it
// hands over the header's fields, the reference lists (planes + per-4x4 motion data of every entry) and the current picture's
// buffers. Scope as the Python module: progressive 8-bit 4:2:0 Baseline / Main / High profile (I / P / B slices, CABAC with
// cabac_init_idc 0 or CAVLC, one slice per picture, 4x4 and -- CABAC only -- 8x8 transform with flat scaling matrices). tests/test_h264_native.py: bit-exact against the Python decoder, picture by
// picture, on all four reference files.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sa_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------- tables
const uint8_t RANGE_LPS[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195}, {111, 135, 160, 185},
    {105, 128, 152, 175}, {100, 122, 144, 166}, {95, 116, 137, 158},  {90, 110, 130, 150},  {85, 104, 123, 142},  {81, 99, 117, 135},
    {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},   {66, 80, 95, 110},    {62, 76, 90, 104},    {59, 72, 86, 99},
    {56, 69, 81, 94},     {53, 65, 77, 89},     {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},
    {41, 50, 59, 69},     {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},
    {30, 37, 43, 50},     {29, 35, 41, 48},     {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},     {23, 28, 33, 39},
    {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},     {18, 22, 26, 30},     {17, 21, 25, 28},
    {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},     {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},
    {12, 14, 17, 20},     {11, 14, 16, 19},     {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},
    {9, 11, 12, 14},      {8, 10, 12, 14},      {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
    {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};
const uint8_t TRANS_LPS[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                               24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};
// (m, n) for ctxIdx 0..275: I slices (Tables 9-12..9-23 column "I") and P / B slices with cabac_init_idc 0
const int8_t CTX_I[276][2] = {
    {20, -15}, {2, 54}, {3, 74}, {20, -15}, {2, 54}, {3, 74}, {-28, 127}, {-23, 104}, {-6, 53}, {-1, 54}, {7, 51},
    // 11..59: not used in I slices
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0},
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0},
    {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0},
    // 60..69
    {0, 41}, {0, 63}, {0, 63}, {0, 63}, {-9, 83}, {4, 86}, {0, 97}, {-7, 72}, {13, 41}, {3, 62},
    // 70..87
    {0, 11}, {1, 55}, {0, 69}, {-17, 127}, {-13, 102}, {0, 82}, {-7, 74}, {-21, 107}, {-27, 127}, {-31, 127}, {-24, 127}, {-18, 95},
    {-27, 127}, {-21, 114}, {-30, 127}, {-17, 123}, {-12, 115}, {-16, 122},
    // 88..104
    {-11, 115}, {-12, 63}, {-2, 68}, {-15, 84}, {-13, 104}, {-3, 70}, {-8, 93}, {-10, 90}, {-30, 127}, {-1, 74}, {-6, 97}, {-7, 91},
    {-20, 127}, {-4, 56}, {-5, 82}, {-7, 76}, {-22, 125},
    // 105..135
    {-7, 93}, {-11, 87}, {-3, 77}, {-5, 71}, {-4, 63}, {-4, 68}, {-12, 84}, {-7, 62}, {-7, 65}, {8, 61}, {5, 56}, {-2, 66}, {1, 64},
    {0, 61}, {-2, 78}, {1, 50}, {7, 52}, {10, 35}, {0, 44}, {11, 38}, {1, 45}, {0, 46}, {5, 44}, {31, 17}, {1, 51}, {7, 50}, {28, 19},
    {16, 33}, {14, 62}, {-13, 108}, {-15, 100},
    // 136..165
    {-13, 101}, {-13, 91}, {-12, 94}, {-10, 88}, {-16, 84}, {-10, 86}, {-7, 83}, {-13, 87}, {-19, 94}, {1, 70}, {0, 72}, {-5, 74},
    {18, 59}, {-8, 102}, {-15, 100}, {0, 95}, {-4, 75}, {2, 72}, {-11, 75}, {-3, 71}, {15, 46}, {-13, 69}, {0, 62}, {0, 65}, {21, 37},
    {-15, 72}, {9, 57}, {16, 54}, {0, 62}, {12, 72},
    // 166..196
    {24, 0}, {15, 9}, {8, 25}, {13, 18}, {15, 9}, {13, 19}, {10, 37}, {12, 18}, {6, 29}, {20, 33}, {15, 30}, {4, 45}, {1, 58}, {0, 62},
    {7, 61}, {12, 38}, {11, 45}, {15, 39}, {11, 42}, {13, 44}, {16, 45}, {12, 41}, {10, 49}, {30, 34}, {18, 42}, {10, 55}, {17, 51},
    {17, 46}, {0, 89}, {26, -19}, {22, -17},
    // 197..226
    {26, -17}, {30, -25}, {28, -20}, {33, -23}, {37, -27}, {33, -23}, {40, -28}, {38, -17}, {33, -11}, {40, -15}, {41, -6}, {38, 1},
    {41, 17}, {30, -6}, {27, 3}, {26, 22}, {37, -16}, {35, -4}, {38, -8}, {38, -3}, {37, 3}, {38, 5}, {42, 0}, {35, 16}, {39, 22},
    {14, 48}, {27, 37}, {21, 60}, {12, 68}, {2, 97},
    // 227..275
    {-3, 71}, {-6, 42}, {-5, 50}, {-3, 54}, {-2, 62}, {0, 58}, {1, 63}, {-2, 72}, {-1, 74}, {-9, 91}, {-5, 67}, {-5, 27}, {-3, 39},
    {-2, 44}, {0, 46}, {-16, 64}, {-8, 68}, {-10, 78}, {-6, 77}, {-10, 86}, {-12, 92}, {-15, 55}, {-10, 60}, {-6, 62}, {-4, 65},
    {-12, 73}, {-8, 76}, {-7, 80}, {-9, 88}, {-17, 110}, {-11, 97}, {-20, 84}, {-11, 79}, {-6, 73}, {-4, 74}, {-13, 86}, {-13, 96},
    {-11, 97}, {-19, 117}, {-8, 78}, {-5, 33}, {-4, 48}, {-2, 53}, {-3, 62}, {-13, 71}, {-10, 79}, {-12, 86}, {-13, 90}, {-14, 97}};
const int8_t CTX_PB0[276][2] = {
    {20, -15}, {2, 54}, {3, 74}, {20, -15}, {2, 54}, {3, 74}, {-28, 127}, {-23, 104}, {-6, 53}, {-1, 54}, {7, 51},
    // 11..23
    {23, 33}, {23, 2}, {21, 0}, {1, 9}, {0, 49}, {-37, 118}, {5, 57}, {-13, 78}, {-11, 65}, {1, 62}, {12, 49}, {-4, 73}, {17, 50},
    // 24..39
    {18, 64}, {9, 43}, {29, 0}, {26, 67}, {16, 90}, {9, 104}, {-46, 127}, {-20, 104}, {1, 67}, {-13, 78}, {-11, 65}, {1, 62}, {-6, 86},
    {-17, 95}, {-6, 61}, {9, 45},
    // 40..53
    {-3, 69}, {-6, 81}, {-11, 96}, {6, 55}, {7, 67}, {-5, 86}, {2, 88}, {0, 58}, {-3, 76}, {-10, 94}, {5, 54}, {4, 69}, {-3, 81}, {0, 88},
    // 54..59
    {-7, 67}, {-5, 74}, {-4, 74}, {-5, 80}, {-7, 72}, {1, 58},
    // 60..69
    {0, 41}, {0, 63}, {0, 63}, {0, 63}, {-9, 83}, {4, 86}, {0, 97}, {-7, 72}, {13, 41}, {3, 62},
    // 70..104
    {0, 45}, {-4, 78}, {-3, 96}, {-27, 126}, {-28, 98}, {-25, 101}, {-23, 67}, {-28, 82}, {-20, 94}, {-16, 83}, {-22, 110}, {-21, 91},
    {-18, 102}, {-13, 93}, {-29, 127}, {-7, 92}, {-5, 89}, {-7, 96}, {-13, 108}, {-3, 46}, {-1, 65}, {-1, 57}, {-9, 93}, {-3, 74},
    {-9, 92}, {-8, 87}, {-23, 126}, {5, 54}, {6, 60}, {6, 59}, {6, 69}, {-1, 48}, {0, 68}, {-4, 69}, {-8, 88},
    // 105..165
    {-2, 85}, {-6, 78}, {-1, 75}, {-7, 77}, {2, 54}, {5, 50}, {-3, 68}, {1, 50}, {6, 42}, {-4, 81}, {1, 63}, {-4, 70}, {0, 67},
    {2, 57}, {-2, 76}, {11, 35}, {4, 64}, {1, 61}, {11, 35}, {18, 25}, {12, 24}, {13, 29}, {13, 36}, {-10, 93}, {-7, 73}, {-2, 73},
    {13, 46}, {9, 49}, {-7, 100}, {9, 53}, {2, 53}, {5, 53}, {-2, 61}, {0, 56}, {0, 56}, {-13, 63}, {-5, 60}, {-1, 62}, {4, 57},
    {-6, 69}, {4, 57}, {14, 39}, {4, 51}, {13, 68}, {3, 64}, {1, 61}, {9, 63}, {7, 50}, {16, 39}, {5, 44}, {4, 52}, {11, 48},
    {-5, 60}, {-1, 59}, {0, 59}, {22, 33}, {5, 44}, {14, 43}, {-1, 78}, {0, 60}, {9, 69},
    // 166..226
    {11, 28}, {2, 40}, {3, 44}, {0, 49}, {0, 46}, {2, 44}, {2, 51}, {0, 47}, {4, 39}, {2, 62}, {6, 46}, {0, 54}, {3, 54}, {2, 58},
    {4, 63}, {6, 51}, {6, 57}, {7, 53}, {6, 52}, {6, 55}, {11, 45}, {14, 36}, {8, 53}, {-1, 82}, {7, 55}, {-3, 78}, {15, 46},
    {22, 31}, {-1, 84}, {25, 7}, {30, -7}, {28, 3}, {28, 4}, {32, 0}, {34, -1}, {30, 6}, {30, 6}, {32, 9}, {31, 19}, {26, 27},
    {26, 30}, {37, 20}, {28, 34}, {17, 70}, {1, 67}, {5, 59}, {9, 67}, {16, 30}, {18, 32}, {18, 35}, {22, 29}, {24, 31}, {23, 38},
    {18, 43}, {20, 41}, {11, 63}, {9, 59}, {9, 64}, {-1, 94}, {-2, 89}, {-9, 108},
    // 227..275
    {-6, 76}, {-2, 44}, {0, 45}, {0, 52}, {-3, 64}, {-2, 59}, {-4, 70}, {-4, 75}, {-8, 82}, {-17, 102}, {-9, 77}, {3, 24}, {0, 42},
    {0, 48}, {0, 55}, {-6, 59}, {-7, 71}, {-12, 83}, {-11, 87}, {-30, 119}, {1, 58}, {-3, 29}, {-1, 36}, {1, 38}, {2, 43}, {-6, 55},
    {0, 58}, {0, 64}, {-3, 74}, {-10, 90}, {0, 70}, {-4, 29}, {5, 31}, {7, 42}, {1, 59}, {-2, 58}, {-3, 72}, {-3, 81}, {-11, 97},
    {0, 58}, {8, 5}, {10, 14}, {14, 18}, {13, 27}, {2, 40}, {0, 58}, {-3, 70}, {-6, 79}, {-8, 85}};

// High profile: ctxIdx 399..435 (transform_size_8x8_flag; 8x8 luma blocks of frame macroblocks: significant_coeff_flag 402..416,
// last_significant_coeff_flag 417..425, coeff_abs_level_minus1 426..435), I slices / cabac_init_idc 0
const int8_t CTX8_I[37][2] = {{31, 21}, {31, 31}, {25, 50}, {-17, 120}, {-20, 112}, {-18, 114}, {-11, 85}, {-15, 92}, {-14, 89}, {-26, 71}, {-15, 81},
                              {-14, 80}, {0, 68}, {-14, 70}, {-24, 56}, {-23, 68}, {-24, 50}, {-11, 74}, {23, -13}, {26, -13}, {40, -15}, {49, -14},
                              {44, 3}, {45, 6}, {44, 34}, {33, 54}, {19, 82}, {-3, 75}, {-1, 23}, {1, 34}, {1, 43}, {0, 54}, {-2, 55}, {0, 61},
                              {1, 64}, {0, 68}, {-9, 92}};
const int8_t CTX8_PB0[37][2] = {{12, 40}, {11, 51}, {14, 59}, {-4, 79}, {-7, 71}, {-5, 69}, {-9, 70}, {-8, 66}, {-10, 68}, {-19, 73}, {-12, 69},
                                {-16, 70}, {-15, 67}, {-20, 62}, {-19, 70}, {-16, 66}, {-22, 65}, {-20, 63}, {9, -2}, {26, -9}, {33, -9}, {39, -7},
                                {41, -2}, {45, 3}, {49, 9}, {45, 27}, {36, 59}, {-6, 66}, {-7, 35}, {-7, 42}, {-8, 45}, {-5, 48}, {-12, 56},
                                {-6, 60}, {-5, 62}, {-8, 66}, {-8, 76}};
const uint8_t SIG8[63] = {0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7,
                          7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8, 6, 11, 12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12};
const uint8_t LAST8[63] = {0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2,
                           3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
const int NORM_ADJUST8[6][6] = {{20, 18, 32, 19, 25, 24}, {22, 19, 35, 21, 28, 26}, {26, 23, 42, 24, 33, 31},
                                {28, 25, 45, 26, 35, 33}, {32, 28, 51, 30, 40, 38}, {36, 32, 58, 34, 46, 43}};
const uint8_t BLK_X[16] = {0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3}, BLK_Y[16] = {0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3};
const uint8_t XY_BLK[4][4] = {{0, 1, 4, 5}, {2, 3, 6, 7}, {8, 9, 12, 13}, {10, 11, 14, 15}};  // [y][x]
const uint8_t ZZ_X[16] = {0, 1, 0, 0, 1, 2, 3, 2, 1, 0, 1, 2, 3, 3, 2, 3}, ZZ_Y[16] = {0, 0, 1, 2, 1, 0, 0, 1, 2, 3, 3, 2, 1, 2, 3, 3};
const int NORM_ADJUST[6][3] = {{10, 16, 13}, {11, 18, 14}, {13, 20, 16}, {14, 23, 18}, {16, 25, 20}, {18, 29, 23}};
const uint8_t QPC[52] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25,
                         26, 27, 28, 29, 29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39};
const int CAT_CBF[5] = {0, 4, 8, 12, 16}, CAT_SIG[5] = {0, 15, 29, 44, 47}, CAT_ABS[5] = {0, 10, 20, 30, 39};
const uint8_t ALPHA[52] = {0, 0, 0, 0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  0,  4,  4,  5,  6,  7,  8,  9,   10,  12,  13,
                           15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255};
const uint8_t BETA[52] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,  0,  0,  0,  0,  0,  2,  2,  2,  3,  3,  3,  3,  4,  4,  4,
                          6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18};
const uint8_t TC0[52][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0},
                            {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 1, 1},
                            {0, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 1, 2}, {1, 2, 3}, {1, 2, 3},
                            {2, 2, 3}, {2, 2, 4}, {2, 3, 4}, {2, 3, 4}, {3, 3, 5}, {3, 4, 6}, {3, 4, 6}, {4, 5, 7}, {4, 5, 8}, {4, 6, 9}, {5, 7, 10},
                            {6, 8, 11}, {6, 8, 13}, {7, 10, 14}, {8, 11, 16}, {9, 12, 18}, {10, 13, 20}, {11, 15, 23}, {13, 17, 25}};
// B mb_type 1..21: partition shape (0 16x16, 1 16x8, 2 8x16), prediction of partitions 0 / 1 (0 L0, 1 L1, 2 Bi)
const int8_t B_MB[22][3] = {{0, 0, 0}, {0, 0, -1}, {0, 1, -1}, {0, 2, -1}, {1, 0, 0}, {2, 0, 0}, {1, 1, 1}, {2, 1, 1}, {1, 0, 1}, {2, 0, 1}, {1, 1, 0},
                            {2, 1, 0}, {1, 0, 2}, {2, 0, 2}, {1, 1, 2}, {2, 1, 2}, {1, 2, 0}, {2, 2, 0}, {1, 2, 1}, {2, 2, 1}, {1, 2, 2}, {2, 2, 2}};
// sub-partition shapes: 0 8x8, 1 8x4, 2 4x8, 3 4x4; B sub_mb_type 1..12 -> (shape, prediction)
const int8_t B_SUB[13][2] = {{-1, -1}, {0, 0}, {0, 1}, {0, 2}, {1, 0}, {2, 0}, {1, 1}, {2, 1}, {1, 2}, {2, 2}, {3, 0}, {3, 1}, {3, 2}};
// CAVLC (Tables 9-5, 9-7 .. 9-10, 9-4), [4 * TotalCoeff + TrailingOnes]
const uint8_t CT_LEN[4][68] = {
    {1, 0, 0, 0, 6, 2, 0, 0, 8, 6, 3, 0, 9, 8, 7, 5, 10, 9, 8, 6, 11, 10, 9, 7, 13, 11, 10, 8, 13, 13, 11, 9, 13, 13, 13, 10, 14, 14, 13, 11,
     14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15, 14, 16, 15, 15, 15, 16, 16, 16, 15, 16, 16, 16, 16, 16, 16, 16, 16},
    {2, 0, 0, 0, 6, 2, 0, 0, 6, 5, 3, 0, 7, 6, 6, 4, 8, 6, 6, 4, 8, 7, 7, 5, 9, 8, 8, 6, 11, 9, 9, 6, 11, 11, 11, 7, 12, 11, 11, 9,
     12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 14, 13, 13, 14, 14, 14, 13, 14, 14, 14, 14},
    {4, 0, 0, 0, 6, 4, 0, 0, 6, 5, 4, 0, 6, 5, 5, 4, 7, 5, 5, 4, 7, 5, 5, 4, 7, 6, 6, 4, 7, 6, 6, 4, 8, 7, 7, 5, 8, 8, 7, 6,
     9, 8, 8, 7, 9, 9, 8, 8, 9, 9, 9, 8, 10, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10},
    {6, 0, 0, 0, 6, 6, 0, 0, 6, 6, 6, 0, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6,
     6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6}};
const uint8_t CT_BITS[4][68] = {
    {1, 0, 0, 0, 5, 1, 0, 0, 7, 4, 1, 0, 7, 6, 5, 3, 7, 6, 5, 3, 7, 6, 5, 4, 15, 6, 5, 4, 11, 14, 5, 4, 8, 10, 13, 4, 15, 14, 9, 4,
     11, 10, 13, 12, 15, 14, 9, 12, 11, 10, 13, 8, 15, 1, 9, 12, 11, 14, 13, 8, 7, 10, 9, 12, 4, 6, 5, 8},
    {3, 0, 0, 0, 11, 2, 0, 0, 7, 7, 3, 0, 7, 10, 9, 5, 7, 6, 5, 4, 4, 6, 5, 6, 7, 6, 5, 8, 15, 6, 5, 4, 11, 14, 13, 4, 15, 10, 9, 4,
     11, 14, 13, 12, 8, 10, 9, 8, 15, 14, 13, 12, 11, 10, 9, 12, 7, 11, 6, 8, 9, 8, 10, 1, 7, 6, 5, 4},
    {15, 0, 0, 0, 15, 14, 0, 0, 11, 15, 13, 0, 8, 12, 14, 12, 15, 10, 11, 11, 11, 8, 9, 10, 9, 14, 13, 9, 8, 10, 9, 8, 15, 14, 13, 13,
     11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9, 12, 8, 10, 13, 8, 13, 7, 9, 12, 9, 12, 11, 10, 5, 8, 7, 6, 1, 4, 3, 2},
    {3,  0,  0,  0,  0,  1,  0,  0,  4,  5,  6,  0,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
     30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63}};
const uint8_t CDC_LEN[20] = {2, 0, 0, 0, 6, 1, 0, 0, 6, 6, 3, 0, 6, 7, 7, 6, 6, 8, 8, 7};
const uint8_t CDC_BITS[20] = {1, 0, 0, 0, 7, 1, 0, 0, 4, 6, 1, 0, 3, 3, 2, 5, 2, 3, 2, 0};
const uint8_t TZ_LEN[15][16] = {{1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9}, {3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6}, {4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6},
                                {5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5},          {4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5},          {6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6},
                                {6, 5, 3, 3, 3, 2, 3, 4, 3, 6},                   {6, 4, 5, 3, 2, 2, 3, 3, 6},                   {6, 6, 4, 2, 2, 3, 2, 5},
                                {5, 5, 3, 2, 2, 2, 4},                            {4, 4, 3, 3, 1, 3},                            {4, 4, 2, 1, 3},
                                {3, 3, 1, 2},                                     {2, 2, 1},                                     {1, 1}};
const uint8_t TZ_BITS[15][16] = {{1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1}, {7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0}, {5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0},
                                 {3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0},          {5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0},          {1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0},
                                 {1, 1, 5, 4, 3, 3, 2, 1, 1, 0},                   {1, 1, 1, 3, 3, 2, 2, 1, 0},                   {1, 0, 1, 3, 2, 1, 1, 1},
                                 {1, 0, 1, 3, 2, 1, 1},                            {0, 1, 1, 2, 1, 3},                            {0, 1, 1, 1, 1},
                                 {0, 1, 1, 1},                                     {0, 1, 1},                                     {0, 1}};
const uint8_t TZ_N[15] = {16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2};
const uint8_t CTZ_LEN[3][4] = {{1, 2, 3, 3}, {1, 2, 2, 0}, {1, 1, 0, 0}}, CTZ_BITS[3][4] = {{1, 1, 1, 0}, {1, 1, 0, 0}, {1, 0, 0, 0}};
const uint8_t RUN_LEN[7][15] = {{1, 1}, {1, 2, 2}, {2, 2, 2, 2}, {2, 2, 2, 3, 3}, {2, 2, 3, 3, 3, 3}, {2, 3, 3, 3, 3, 3, 3}, {3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11}};
const uint8_t RUN_BITS[7][15] = {{1, 0}, {1, 1, 0}, {3, 2, 1, 0}, {3, 2, 1, 1, 0}, {3, 2, 3, 2, 1, 0}, {3, 0, 1, 3, 2, 5, 4}, {7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1}};
const uint8_t RUN_N[7] = {2, 3, 4, 5, 6, 7, 15};
const uint8_t CBP_INTRA[48] = {47, 31, 15, 0,  23, 27, 29, 30, 7,  11, 13, 14, 39, 43, 45, 46, 16, 3,  5,  10, 12, 19, 21, 26,
                               28, 35, 37, 42, 44, 1,  2,  4,  8,  17, 18, 20, 24, 6,  9,  22, 25, 32, 33, 34, 36, 40, 38, 41};
const uint8_t CBP_INTER[48] = {0,  16, 1,  2,  4,  8,  32, 3,  5,  10, 12, 15, 47, 7,  11, 13, 14, 6,  9,  31, 35, 37, 42, 44,
                               33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41};

inline int level_scale8(int qp, int i, int j) {
  const int* v = NORM_ADJUST8[qp % 6];
  int k;
  if (i % 4 == 0 && j % 4 == 0) k = 0;
  else if (i % 2 == 1 && j % 2 == 1) k = 1;
  else if (i % 4 == 2 && j % 4 == 2) k = 2;
  else if ((i % 4 == 0 && j % 2 == 1) || (i % 2 == 1 && j % 4 == 0)) k = 3;
  else if ((i % 4 == 0 && j % 4 == 2) || (i % 4 == 2 && j % 4 == 0)) k = 4;
  else k = 5;
  return 16 * v[k];
}
inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
inline int clip1(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
inline int level_scale(int qp, int x, int y) {
  const int* v = NORM_ADJUST[qp % 6];
  return 16 * ((x % 2 == 0 && y % 2 == 0) ? v[0] : ((x % 2 == 1 && y % 2 == 1) ? v[1] : v[2]));
}

struct Desync {
  const char* what;
};
#define H264_CHECK(cond, msg)      \
  do {                             \
    if (!(cond)) throw Desync{msg}; \
  } while (0)

struct Bits {
  const uint8_t* d;
  int64_t n_bits, p = 0;
  int u1() {
    if (p >= n_bits) {  // (zero padding behind the payload: the arithmetic decoder reads a few bits ahead)
      H264_CHECK(p < n_bits + 64, "read beyond the slice data");
      ++p;
      return 0;
    }
    const int v = (d[p >> 3] >> (7 - (p & 7))) & 1;
    ++p;
    return v;
  }
  int u(int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) v = (v << 1) | u1();
    return v;
  }
  int ue() {
    int z = 0;
    while (u1() == 0) {
      ++z;
      H264_CHECK(z < 32, "Exp-Golomb prefix runaway");
    }
    return (1 << z) - 1 + (z ? u(z) : 0);
  }
  int se() {
    const int k = ue();
    return (k & 1) ? (k + 1) / 2 : -(k / 2);
  }
};

struct Cabac {
  Bits* b;
  int range = 510, offset = 0;
  uint8_t state[436], mps[436];
  void init(Bits* bits, int qp, const int8_t (*tab)[2], const int8_t (*tab8)[2]) {
    b = bits;
    offset = b->u(9);
    const int q = clip3(0, 51, qp);
    for (int k = 0; k < 436; ++k) {
      if (k >= 276 && k < 399) {
        state[k] = 0, mps[k] = 0;
        continue;
      }
      const int8_t* mn = k < 276 ? tab[k] : tab8[k - 399];
      const int pre = clip3(1, 126, ((mn[0] * q) >> 4) + mn[1]);
      if (pre <= 63) {
        state[k] = (uint8_t)(63 - pre);
        mps[k] = 0;
      } else {
        state[k] = (uint8_t)(pre - 64);
        mps[k] = 1;
      }
    }
  }
  void renorm() {
    while (range < 256) {
      range <<= 1;
      offset = (offset << 1) | b->u1();
    }
  }
  int decision(int ctx) {
    const int s = state[ctx];
    const int lps = RANGE_LPS[s][(range >> 6) & 3];
    int v;
    range -= lps;
    if (offset >= range) {
      v = 1 - mps[ctx];
      offset -= range;
      range = lps;
      if (s == 0) mps[ctx] = 1 - mps[ctx];
      state[ctx] = TRANS_LPS[s];
    } else {
      v = mps[ctx];
      state[ctx] = (uint8_t)std::min(s + 1, 62);
    }
    renorm();
    return v;
  }
  int bypass() {
    offset = (offset << 1) | b->u1();
    if (offset >= range) {
      offset -= range;
      return 1;
    }
    return 0;
  }
  int terminate() {
    range -= 2;
    if (offset >= range) return 1;
    renorm();
    return 0;
  }
};

enum { T_NONE = 0, T_I4 = 1, T_I16 = 2, T_INTER = 3, T_I8 = 4 };  // T_I4 and T_I8 are both mb_type I_NxN
struct MB {
  uint8_t typ = T_NONE, skip = 0, direct16 = 0, intra = 0, ref0 = 0, t8 = 0, i16 = 0, cbp_luma = 0, cbp_chroma = 0, chroma_mode = 0, cbf_dc = 0, qp_delta_nz = 0;
  int8_t qp = 0;
  uint8_t modes[16], cbf_luma[16], cbf_cdc[2], cbf_cac[2][4];
  MB() {
    for (int i = 0; i < 16; ++i) modes[i] = 2, cbf_luma[i] = 0;
    cbf_cdc[0] = cbf_cdc[1] = 0;
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < 4; ++i) cbf_cac[c][i] = 0;
  }
};

struct Nb {
  bool avail;
  int ref;
  int mvx, mvy;
};
struct Part {
  int sx, sy, w, h, pred, shape, pi, g;
};

struct Slice {
  const sa_h264_slice* s;
  const sa_h264_pic *l0, *l1;
  sa_h264_pic* cur;
  Bits bits;
  Cabac cab;
  int W, Hh, W4, H4;
  int stype, qp, prev_qp_delta_nz = 0;
  std::vector<MB> mbs;
  std::vector<int16_t> mvd;      // [2][H4][W4][2]
  std::vector<uint8_t> direct, done, nz;
  std::vector<int32_t> tc, tcc[2];
  std::vector<int> implicit;     // [n0][n1][2]
  int64_t stop_bit = 0;
  int cur_mx = 0, cur_my = 0;
  int cur_dx4 = 0, cur_dy4 = 0;  // 4x4 position of the macroblock direct_parts looks at
  int stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // I4, I16, skip, inter, bits_left

  // ---- accessors
  const sa_h264_pic* list(int l) const { return l ? l1 : l0; }
  int16_t* MV(int l, int y4, int x4) { return cur->mv + (((size_t)l * H4 + y4) * W4 + x4) * 2; }
  int8_t& REF(int l, int y4, int x4) { return cur->ref[((size_t)l * H4 + y4) * W4 + x4]; }
  int32_t& REFID(int l, int y4, int x4) { return cur->refid[((size_t)l * H4 + y4) * W4 + x4]; }
  int16_t* MVD(int l, int y4, int x4) { return mvd.data() + (((size_t)l * H4 + y4) * W4 + x4) * 2; }
  MB* mb(int mx, int my) {
    if (mx < 0 || my < 0 || mx >= W || my >= Hh) return nullptr;
    MB* m = &mbs[(size_t)my * W + mx];
    return m->typ == T_NONE ? nullptr : m;
  }
  uint8_t& Y(int y, int x) { return cur->y[(size_t)y * (W * 16) + x]; }
  uint8_t* C(int c) { return c ? cur->cr : cur->cb; }

  // ---- neighbour motion data (8.4.1.3)
  Nb nb(int l, int x4, int y4) {
    if (x4 < 0 || y4 < 0 || x4 >= W4 || y4 >= H4 || !done[(size_t)y4 * W4 + x4]) return {false, -1, 0, 0};
    const int rf = REF(l, y4, x4);
    if (rf < 0) return {true, -1, 0, 0};
    const int16_t* m = MV(l, y4, x4);
    return {true, rf, m[0], m[1]};
  }
  static int med(int a, int b, int c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
  void mvp(int l, int x4, int y4, int w4, int ref, int shape, int part, int& px, int& py) {
    Nb a = nb(l, x4 - 1, y4), b = nb(l, x4, y4 - 1), c = nb(l, x4 + w4, y4 - 1);
    if (!c.avail) c = nb(l, x4 - 1, y4 - 1);
    if (shape == 1) {  // 16x8
      if (part == 0 && b.ref == ref) { px = b.mvx, py = b.mvy; return; }
      if (part == 1 && a.ref == ref) { px = a.mvx, py = a.mvy; return; }
    } else if (shape == 2) {  // 8x16
      if (part == 0 && a.ref == ref) { px = a.mvx, py = a.mvy; return; }
      if (part == 1 && c.ref == ref) { px = c.mvx, py = c.mvy; return; }
    }
    if (!b.avail && !c.avail && a.avail) b = c = a;
    const int n = (a.ref == ref) + (b.ref == ref) + (c.ref == ref);
    if (n == 1) {
      const Nb& k = a.ref == ref ? a : (b.ref == ref ? b : c);
      px = k.mvx, py = k.mvy;
      return;
    }
    px = med(a.mvx, b.mvx, c.mvx), py = med(a.mvy, b.mvy, c.mvy);
  }
  void set_motion(int l, int x4, int y4, int w4, int h4, int ref, int mvx, int mvy, int dx = 0, int dy = 0) {
    int32_t id = -1;
    if (ref >= 0) {
      H264_CHECK(ref < s->nref[l] && list(l)[ref].y, "reference index names an empty list entry");
      id = list(l)[ref].id;
    }
    for (int y = y4; y < y4 + h4; ++y)
      for (int x = x4; x < x4 + w4; ++x) {
        REF(l, y, x) = (int8_t)ref;
        REFID(l, y, x) = id;
        int16_t* m = MV(l, y, x);
        m[0] = (int16_t)mvx, m[1] = (int16_t)mvy;
        int16_t* d = MVD(l, y, x);
        d[0] = (int16_t)std::min(std::abs(dx), 32767), d[1] = (int16_t)std::min(std::abs(dy), 32767);
      }
  }

  // ---- direct prediction (8.4.1.2)
  void col(int x4, int y4, bool& intra, int32_t& refid, int& mvx, int& mvy, int& idx) {
    const sa_h264_pic& c = l1[0];
    intra = true, refid = -1, mvx = mvy = 0, idx = -1;
    if (c.intra4[(size_t)y4 * W4 + x4]) return;
    for (int l = 0; l < 2; ++l) {
      const int r = c.ref[((size_t)l * H4 + y4) * W4 + x4];
      if (r >= 0) {
        intra = false;
        refid = c.refid[((size_t)l * H4 + y4) * W4 + x4];
        const int16_t* m = c.mv + (((size_t)l * H4 + y4) * W4 + x4) * 2;
        mvx = m[0], mvy = m[1], idx = r;
        return;
      }
    }
  }
  void direct_pred(int mx, int my, const int* quads, int nq) {
    const int X4 = mx * 4, Y4 = my * 4;
    const bool inf8 = s->direct_8x8_inference != 0;
    H264_CHECK(s->nref[1] >= 1 && l1[0].y, "direct prediction without RefPicList1[0]");
    int refs[2] = {-1, -1}, mvs[2][2] = {{0, 0}, {0, 0}};
    bool zero_pred = false;
    if (s->direct_spatial) {
      for (int l = 0; l < 2; ++l) {
        Nb a = nb(l, X4 - 1, Y4), b = nb(l, X4, Y4 - 1), c = nb(l, X4 + 4, Y4 - 1);
        if (!c.avail) c = nb(l, X4 - 1, Y4 - 1);
        int rr = -1;
        for (const Nb* n : {&a, &b, &c})
          if (n->ref >= 0 && (rr < 0 || n->ref < rr)) rr = n->ref;
        refs[l] = rr;
      }
      zero_pred = refs[0] < 0 && refs[1] < 0;
      if (zero_pred)
        refs[0] = refs[1] = 0;
      else
        for (int l = 0; l < 2; ++l)
          if (refs[l] >= 0) mvp(l, X4, Y4, 4, refs[l], 0, 0, mvs[l][0], mvs[l][1]);
    }
    for (int qi = 0; qi < nq; ++qi) {
      const int q = quads[qi], qx = (q & 1) * 2, qy = (q >> 1) * 2;
      const int nb_ = inf8 ? 1 : 4;
      for (int k = 0; k < nb_; ++k) {
        const int bx = inf8 ? qx : qx + (k & 1), by = inf8 ? qy : qy + (k >> 1), w = inf8 ? 2 : 1;
        const int cx = inf8 ? (qx * 3) / 2 : bx, cy = inf8 ? (qy * 3) / 2 : by;
        bool cintra;
        int32_t cref;
        int cmx, cmy, cidx;
        col(X4 + cx, Y4 + cy, cintra, cref, cmx, cmy, cidx);
        if (s->direct_spatial) {
          const bool col_zero = !cintra && cidx == 0 && std::abs(cmx) <= 1 && std::abs(cmy) <= 1;
          for (int l = 0; l < 2; ++l) {
            int mvx = mvs[l][0], mvy = mvs[l][1];
            if (zero_pred || refs[l] < 0 || (refs[l] == 0 && col_zero)) mvx = mvy = 0;
            set_motion(l, X4 + bx, Y4 + by, w, w, refs[l], mvx, mvy);
          }
        } else {
          int r0 = 0;
          if (!cintra) {
            r0 = -1;
            for (int i = 0; i < s->nref[0]; ++i)
              if (l0[i].y && l0[i].id == cref) {
                r0 = i;
                break;
              }
            H264_CHECK(r0 >= 0, "temporal direct: the co-located block's reference is not in RefPicList0");
          }
          H264_CHECK(l0[r0].y, "temporal direct: empty RefPicList0 entry");
          const int tb = clip3(-128, 127, cur->poc - l0[r0].poc), td = clip3(-128, 127, l1[0].poc - l0[r0].poc);
          int m0x, m0y, m1x, m1y;
          if (td == 0) {
            m0x = cmx, m0y = cmy, m1x = m1y = 0;
          } else {
            const int tx = (16384 + std::abs(td / 2)) / td;
            const int dsf = clip3(-1024, 1023, (tb * tx + 32) >> 6);
            m0x = (dsf * cmx + 128) >> 8, m0y = (dsf * cmy + 128) >> 8;
            m1x = m0x - cmx, m1y = m0y - cmy;
          }
          set_motion(0, X4 + bx, Y4 + by, w, w, r0, m0x, m0y);
          set_motion(1, X4 + bx, Y4 + by, w, w, 0, m1x, m1y);
        }
        for (int y = by; y < by + w; ++y)
          for (int x = bx; x < bx + w; ++x) direct[(size_t)(Y4 + y) * W4 + X4 + x] = 1;
      }
    }
  }

  // ---- interpolation (8.4.2.2)
  static inline int S(const uint8_t* p, int w, int h, int x, int y) { return p[(size_t)clip3(0, h - 1, y) * w + clip3(0, w - 1, x)]; }
  static int luma_sample(const uint8_t* p, int w, int h, int xq, int yq) {
    const int xi = xq >> 2, yi = yq >> 2, fx = xq & 3, fy = yq & 3;
    auto b1 = [&](int x, int y) { return S(p, w, h, x - 2, y) - 5 * S(p, w, h, x - 1, y) + 20 * S(p, w, h, x, y) + 20 * S(p, w, h, x + 1, y) - 5 * S(p, w, h, x + 2, y) + S(p, w, h, x + 3, y); };
    auto h1 = [&](int x, int y) { return S(p, w, h, x, y - 2) - 5 * S(p, w, h, x, y - 1) + 20 * S(p, w, h, x, y) + 20 * S(p, w, h, x, y + 1) - 5 * S(p, w, h, x, y + 2) + S(p, w, h, x, y + 3); };
    const int G = S(p, w, h, xi, yi);
    if (fx == 0 && fy == 0) return G;
    if (fy == 0) {
      const int b = clip1((b1(xi, yi) + 16) >> 5);
      return fx == 2 ? b : ((b + (fx == 1 ? G : S(p, w, h, xi + 1, yi)) + 1) >> 1);
    }
    if (fx == 0) {
      const int hh = clip1((h1(xi, yi) + 16) >> 5);
      return fy == 2 ? hh : ((hh + (fy == 1 ? G : S(p, w, h, xi, yi + 1)) + 1) >> 1);
    }
    if (fx == 2 || fy == 2) {
      const int j1 = b1(xi, yi - 2) - 5 * b1(xi, yi - 1) + 20 * b1(xi, yi) + 20 * b1(xi, yi + 1) - 5 * b1(xi, yi + 2) + b1(xi, yi + 3);
      const int j = clip1((j1 + 512) >> 10);
      if (fx == 2 && fy == 2) return j;
      int o;
      if (fx == 2)
        o = clip1((b1(xi, fy == 1 ? yi : yi + 1) + 16) >> 5);
      else
        o = clip1((h1(fx == 1 ? xi : xi + 1, yi) + 16) >> 5);
      return (j + o + 1) >> 1;
    }
    const int bb = clip1((b1(xi, fy == 1 ? yi : yi + 1) + 16) >> 5), hh = clip1((h1(fx == 1 ? xi : xi + 1, yi) + 16) >> 5);
    return (bb + hh + 1) >> 1;
  }
  static int chroma_sample(const uint8_t* p, int w, int h, int x8, int y8) {
    const int xi = x8 >> 3, yi = y8 >> 3, fx = x8 & 7, fy = y8 & 7;
    return ((8 - fx) * (8 - fy) * S(p, w, h, xi, yi) + fx * (8 - fy) * S(p, w, h, xi + 1, yi) + (8 - fx) * fy * S(p, w, h, xi, yi + 1) +
            fx * fy * S(p, w, h, xi + 1, yi + 1) + 32) >> 6;
  }

  // Block forms of the two sample functions above (the same arithmetic; tests/test_h264_native.py compares whole pictures with the
  // Python decoder): the (w + 5) x (h + 5) source window is gathered once -- rows copied when it lies inside the plane, clamped
  // per sample at the picture border -- and the half-sample intermediates are computed per row / column instead of per sample.
  static void gather(const uint8_t* p, int W_, int H_, int x0, int y0, int w, int h, int* R) {
    if (x0 >= 0 && y0 >= 0 && x0 + w <= W_ && y0 + h <= H_) {
      for (int j = 0; j < h; ++j) {
        const uint8_t* r = p + (size_t)(y0 + j) * W_ + x0;
        for (int i = 0; i < w; ++i) R[j * w + i] = r[i];
      }
    } else {
      for (int j = 0; j < h; ++j) {
        const uint8_t* r = p + (size_t)clip3(0, H_ - 1, y0 + j) * W_;
        for (int i = 0; i < w; ++i) R[j * w + i] = r[clip3(0, W_ - 1, x0 + i)];
      }
    }
  }
  static void luma_block(const uint8_t* p, int W_, int H_, int xq, int yq, int w, int h, int* out) {
    const int xi = xq >> 2, yi = yq >> 2, fx = xq & 3, fy = yq & 3;
    if (fx == 0 && fy == 0) {
      gather(p, W_, H_, xi, yi, w, h, out);
      return;
    }
    const int RW = w + 5, RH = h + 5;
    int R[21 * 21], B1[21 * 16], H1[16 * 21];
    gather(p, W_, H_, xi - 2, yi - 2, RW, RH, R);  // R[r][c] = sample (xi - 2 + c, yi - 2 + r)
    auto Rv = [&](int r, int c) { return R[r * RW + c]; };
    const bool need_b = fy == 0 || fx == 2 || (fx != 0 && fy != 0);  // horizontal intermediates b1[r][c]: between columns xi + c, xi + c + 1, row yi - 2 + r
    const bool need_h = fx == 0 || fy == 2 || (fx != 0 && fy != 0);  // vertical intermediates h1[r][c]: between rows yi + r, yi + r + 1, column xi - 2 + c
    if (need_b || fx == 2 || fy == 2)
      for (int r = 0; r < RH; ++r)
        for (int c = 0; c < w; ++c) B1[r * 16 + c] = Rv(r, c) - 5 * Rv(r, c + 1) + 20 * Rv(r, c + 2) + 20 * Rv(r, c + 3) - 5 * Rv(r, c + 4) + Rv(r, c + 5);
    if (need_h)
      for (int r = 0; r < h; ++r)
        for (int c = 0; c < RW; ++c) H1[r * 21 + c] = Rv(r, c) - 5 * Rv(r + 1, c) + 20 * Rv(r + 2, c) + 20 * Rv(r + 3, c) - 5 * Rv(r + 4, c) + Rv(r + 5, c);
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i) {
        const int G = Rv(j + 2, i + 2);
        int v;
        if (fy == 0) {
          const int b = clip1((B1[(j + 2) * 16 + i] + 16) >> 5);
          v = fx == 2 ? b : ((b + (fx == 1 ? G : Rv(j + 2, i + 3)) + 1) >> 1);
        } else if (fx == 0) {
          const int hh = clip1((H1[j * 21 + i + 2] + 16) >> 5);
          v = fy == 2 ? hh : ((hh + (fy == 1 ? G : Rv(j + 3, i + 2)) + 1) >> 1);
        } else if (fx == 2 || fy == 2) {
          const int j1 = B1[j * 16 + i] - 5 * B1[(j + 1) * 16 + i] + 20 * B1[(j + 2) * 16 + i] + 20 * B1[(j + 3) * 16 + i] - 5 * B1[(j + 4) * 16 + i] + B1[(j + 5) * 16 + i];
          const int jj = clip1((j1 + 512) >> 10);
          if (fx == 2 && fy == 2) {
            v = jj;
          } else {
            const int o = fx == 2 ? clip1((B1[(j + (fy == 1 ? 2 : 3)) * 16 + i] + 16) >> 5) : clip1((H1[j * 21 + i + (fx == 1 ? 2 : 3)] + 16) >> 5);
            v = (jj + o + 1) >> 1;
          }
        } else {
          const int bb = clip1((B1[(j + (fy == 1 ? 2 : 3)) * 16 + i] + 16) >> 5), hh = clip1((H1[j * 21 + i + (fx == 1 ? 2 : 3)] + 16) >> 5);
          v = (bb + hh + 1) >> 1;
        }
        out[j * w + i] = v;
      }
  }
  static void chroma_block(const uint8_t* p, int W_, int H_, int x8, int y8, int w, int h, int* out) {
    const int xi = x8 >> 3, yi = y8 >> 3, fx = x8 & 7, fy = y8 & 7;
    if (fx == 0 && fy == 0) {
      gather(p, W_, H_, xi, yi, w, h, out);
      return;
    }
    int R[9 * 9];
    gather(p, W_, H_, xi, yi, w + 1, h + 1, R);
    const int RW = w + 1, w00 = (8 - fx) * (8 - fy), w10 = fx * (8 - fy), w01 = (8 - fx) * fy, w11 = fx * fy;
    for (int j = 0; j < h; ++j)
      for (int i = 0; i < w; ++i)
        out[j * w + i] = (w00 * R[j * RW + i] + w10 * R[j * RW + i + 1] + w01 * R[(j + 1) * RW + i] + w11 * R[(j + 1) * RW + i + 1] + 32) >> 6;
  }
  void predict_inter(int mx, int my, const Part* parts, int np) {
    const int X4 = mx * 4, Y4 = my * 4, LW = W * 16, LH = Hh * 16, CW = W * 8, CH = Hh * 8;
    const int mode = s->weighted_mode;
    int py[2][256], pc[2][2][64];
    for (int k = 0; k < np; ++k) {
      const int x4 = X4 + parts[k].sx, y4 = Y4 + parts[k].sy, w = parts[k].w * 4, hh = parts[k].h * 4;
      int n = 0, rl[2], rr[2];
      // the plain copy (one reference, whole-sample vector, no weighting, window inside the picture: most of a static scene's
      // skipped macroblocks) goes row by row from plane to plane
      {
        const int r0 = REF(0, y4, x4), r1 = REF(1, y4, x4);
        if ((r0 >= 0) != (r1 >= 0)) {
          const int l = r0 >= 0 ? 0 : 1, rf = l ? r1 : r0;
          const sa_h264_pic& rp = list(l)[rf];
          H264_CHECK(rp.y, "prediction from an empty reference list entry");
          const int mvx = MV(l, y4, x4)[0], mvy = MV(l, y4, x4)[1];
          bool plain = mode != 1;
          if (mode == 1) {
            const int32_t(*wt)[2] = s->weights[l][rf];
            plain = wt[0][0] == (1 << s->luma_log2_denom) && wt[0][1] == 0 && wt[1][0] == (1 << s->chroma_log2_denom) && wt[1][1] == 0 &&
                    wt[2][0] == (1 << s->chroma_log2_denom) && wt[2][1] == 0;
          }
          const int sx = x4 * 4 + (mvx >> 2), sy = y4 * 4 + (mvy >> 2);
          if (plain && !(mvx & 7) && !(mvy & 7) && sx >= 0 && sy >= 0 && sx + w <= LW && sy + hh <= LH) {
            for (int j = 0; j < hh; ++j) memcpy(cur->y + (size_t)(y4 * 4 + j) * LW + x4 * 4, rp.y + (size_t)(sy + j) * LW + sx, (size_t)w);
            for (int c = 0; c < 2; ++c) {
              const uint8_t* src = c ? rp.cr : rp.cb;
              uint8_t* dst = C(c);
              for (int j = 0; j < hh / 2; ++j)
                memcpy(dst + (size_t)(y4 * 2 + j) * CW + x4 * 2, src + (size_t)(sy / 2 + j) * CW + sx / 2, (size_t)(w / 2));
            }
            ++stats[7];
            continue;
          }
        }
      }
      // the plain average (two references, whole-sample vectors, default weighting or implicit weights 32 / 32 -- the same
      // arithmetic -- and both windows inside the picture: the static background of B pictures) also goes plane to plane
      {
        const int r0 = REF(0, y4, x4), r1 = REF(1, y4, x4);
        if (r0 >= 0 && r1 >= 0 && mode != 1) {
          const sa_h264_pic &p0 = l0[r0], &p1 = l1[r1];
          H264_CHECK(p0.y && p1.y, "prediction from an empty reference list entry");
          const int m0x = MV(0, y4, x4)[0], m0y = MV(0, y4, x4)[1], m1x = MV(1, y4, x4)[0], m1y = MV(1, y4, x4)[1];
          bool avg = mode == 0;
          if (mode == 2) {
            const int* iw = &implicit[((size_t)r0 * s->nref[1] + r1) * 2];
            avg = iw[0] == 32 && iw[1] == 32;
          }
          const int ax = x4 * 4 + (m0x >> 2), ay = y4 * 4 + (m0y >> 2), bx = x4 * 4 + (m1x >> 2), by = y4 * 4 + (m1y >> 2);
          if (avg && !((m0x | m0y | m1x | m1y) & 7) && ax >= 0 && ay >= 0 && ax + w <= LW && ay + hh <= LH && bx >= 0 && by >= 0 && bx + w <= LW &&
              by + hh <= LH) {
            for (int j = 0; j < hh; ++j) {
              const uint8_t *a = p0.y + (size_t)(ay + j) * LW + ax, *b = p1.y + (size_t)(by + j) * LW + bx;
              uint8_t* d = cur->y + (size_t)(y4 * 4 + j) * LW + x4 * 4;
              for (int i = 0; i < w; ++i) d[i] = (uint8_t)((a[i] + b[i] + 1) >> 1);
            }
            for (int c = 0; c < 2; ++c) {
              const uint8_t *sa_ = c ? p0.cr : p0.cb, *sb_ = c ? p1.cr : p1.cb;
              uint8_t* dst = C(c);
              for (int j = 0; j < hh / 2; ++j) {
                const uint8_t *a = sa_ + (size_t)(ay / 2 + j) * CW + ax / 2, *b = sb_ + (size_t)(by / 2 + j) * CW + bx / 2;
                uint8_t* d = dst + (size_t)(y4 * 2 + j) * CW + x4 * 2;
                for (int i = 0; i < w / 2; ++i) d[i] = (uint8_t)((a[i] + b[i] + 1) >> 1);
              }
            }
            ++stats[7];
            continue;
          }
        }
      }
      for (int l = 0; l < 2; ++l) {
        const int rf = REF(l, y4, x4);
        if (rf < 0) continue;
        const sa_h264_pic& rp = list(l)[rf];
        H264_CHECK(rp.y, "prediction from an empty reference list entry");
        const int mvx = MV(l, y4, x4)[0], mvy = MV(l, y4, x4)[1];
        luma_block(rp.y, LW, LH, x4 * 16 + mvx, y4 * 16 + mvy, w, hh, py[n]);
        for (int c = 0; c < 2; ++c) chroma_block(c ? rp.cr : rp.cb, CW, CH, x4 * 16 + mvx, y4 * 16 + mvy, w / 2, hh / 2, pc[n][c]);
        rl[n] = l, rr[n] = rf;
        ++n;
      }
      H264_CHECK(n > 0, "inter partition without a reference");
      const int ld = s->luma_log2_denom, cd = s->chroma_log2_denom;
      for (int comp = 0; comp < 3; ++comp) {
        const int bw = comp ? w / 2 : w, bh = comp ? hh / 2 : hh, stride = comp ? CW : LW;
        uint8_t* dst = (comp == 0 ? cur->y : C(comp - 1)) + (size_t)(comp ? y4 * 2 : y4 * 4) * stride + (comp ? x4 * 2 : x4 * 4);
        const int* a = comp ? pc[0][comp - 1] : py[0];
        const int* b = comp ? pc[1][comp - 1] : py[1];
        const int dn = comp ? cd : ld;
        // (one loop per weighting form: the form is the same for every sample of the block)
        if (n == 1 && mode != 1) {
          for (int j = 0; j < bh; ++j)
            for (int i = 0; i < bw; ++i) dst[(size_t)j * stride + i] = (uint8_t)a[j * bw + i];
        } else if (n == 1) {
          const int32_t* wt = s->weights[rl[0]][rr[0]][comp];
          const int w0 = wt[0], o0 = wt[1], rnd = dn >= 1 ? 1 << (dn - 1) : 0;
          for (int j = 0; j < bh; ++j)
            for (int i = 0; i < bw; ++i) dst[(size_t)j * stride + i] = (uint8_t)clip1((dn >= 1 ? ((a[j * bw + i] * w0 + rnd) >> dn) : a[j * bw + i] * w0) + o0);
        } else if (mode == 1) {
          const int32_t *w0 = s->weights[0][rr[0]][comp], *w1 = s->weights[1][rr[1]][comp];
          const int wa = w0[0], wb = w1[0], off = (w0[1] + w1[1] + 1) >> 1;
          for (int j = 0; j < bh; ++j)
            for (int i = 0; i < bw; ++i) dst[(size_t)j * stride + i] = (uint8_t)clip1(((a[j * bw + i] * wa + b[j * bw + i] * wb + (1 << dn)) >> (dn + 1)) + off);
        } else if (mode == 2) {
          const int* iw = &implicit[((size_t)rr[0] * s->nref[1] + rr[1]) * 2];
          const int wa = iw[0], wb = iw[1];
          for (int j = 0; j < bh; ++j)
            for (int i = 0; i < bw; ++i) dst[(size_t)j * stride + i] = (uint8_t)clip1((a[j * bw + i] * wa + b[j * bw + i] * wb + 32) >> 6);
        } else {
          for (int j = 0; j < bh; ++j)
            for (int i = 0; i < bw; ++i) dst[(size_t)j * stride + i] = (uint8_t)((a[j * bw + i] + b[j * bw + i] + 1) >> 1);
        }
      }
    }
  }

  // ---- intra prediction (8.3)
  void pred4(const MB& m, int mx, int my, int blk) {
    const int bx = BLK_X[blk], by = BLK_Y[blk], x0 = mx * 16 + bx * 4, y0 = my * 16 + by * 4;
    const bool left = bx > 0 || mx > 0, top = by > 0 || my > 0;
    bool tr;
    if (by == 0)
      tr = bx == 3 ? (my > 0 && mx + 1 < W) : top;
    else
      tr = bx < 3 && XY_BLK[by - 1][bx + 1] < blk;
    int T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, L[4] = {0, 0, 0, 0}, tl = 0;
    if (top) {
      for (int i = 0; i < 4; ++i) T[i] = Y(y0 - 1, x0 + i);
      for (int i = 4; i < 8; ++i) T[i] = tr ? Y(y0 - 1, x0 + i) : T[3];
    }
    if (left)
      for (int j = 0; j < 4; ++j) L[j] = Y(y0 + j, x0 - 1);
    if ((bx > 0 || mx > 0) && (by > 0 || my > 0)) tl = Y(y0 - 1, x0 - 1);
    auto P = [&](int x, int y) -> int { return y < 0 ? (x < 0 ? tl : T[x]) : L[y]; };
    int out[4][4];
    const int mode = m.modes[blk];
    for (int y = 0; y < 4; ++y)
      for (int x = 0; x < 4; ++x) {
        int v = 0;
        switch (mode) {
          case 0: v = P(x, -1); break;
          case 1: v = P(-1, y); break;
          case 2:
            if (top && left) v = (T[0] + T[1] + T[2] + T[3] + L[0] + L[1] + L[2] + L[3] + 4) >> 3;
            else if (left) v = (L[0] + L[1] + L[2] + L[3] + 2) >> 2;
            else if (top) v = (T[0] + T[1] + T[2] + T[3] + 2) >> 2;
            else v = 128;
            break;
          case 3:
            v = (x == 3 && y == 3) ? (P(6, -1) + 3 * P(7, -1) + 2) >> 2 : (P(x + y, -1) + 2 * P(x + y + 1, -1) + P(x + y + 2, -1) + 2) >> 2;
            break;
          case 4:
            if (x > y) v = (P(x - y - 2, -1) + 2 * P(x - y - 1, -1) + P(x - y, -1) + 2) >> 2;
            else if (x < y) v = (P(-1, y - x - 2) + 2 * P(-1, y - x - 1) + P(-1, y - x) + 2) >> 2;
            else v = (P(0, -1) + 2 * P(-1, -1) + P(-1, 0) + 2) >> 2;
            break;
          case 5: {
            const int z = 2 * x - y;
            if (z >= 0 && z % 2 == 0) v = (P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 1) >> 1;
            else if (z >= 0) v = (P(x - (y >> 1) - 2, -1) + 2 * P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 2) >> 2;
            else if (z == -1) v = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
            else v = (P(-1, y - 1) + 2 * P(-1, y - 2) + P(-1, y - 3) + 2) >> 2;
            break;
          }
          case 6: {
            const int z = 2 * y - x;
            if (z >= 0 && z % 2 == 0) v = (P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 1) >> 1;
            else if (z >= 0) v = (P(-1, y - (x >> 1) - 2) + 2 * P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 2) >> 2;
            else if (z == -1) v = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
            else v = (P(x - 1, -1) + 2 * P(x - 2, -1) + P(x - 3, -1) + 2) >> 2;
            break;
          }
          case 7:
            v = (y % 2 == 0) ? (P(x + (y >> 1), -1) + P(x + (y >> 1) + 1, -1) + 1) >> 1
                             : (P(x + (y >> 1), -1) + 2 * P(x + (y >> 1) + 1, -1) + P(x + (y >> 1) + 2, -1) + 2) >> 2;
            break;
          default: {
            const int z = x + 2 * y;
            if (z > 5) v = P(-1, 3);
            else if (z == 5) v = (P(-1, 2) + 3 * P(-1, 3) + 2) >> 2;
            else if (z % 2 == 0) v = (P(-1, y + (x >> 1)) + P(-1, y + (x >> 1) + 1) + 1) >> 1;
            else v = (P(-1, y + (x >> 1)) + 2 * P(-1, y + (x >> 1) + 1) + P(-1, y + (x >> 1) + 2) + 2) >> 2;
          }
        }
        out[y][x] = v;
      }
    for (int y = 0; y < 4; ++y)
      for (int x = 0; x < 4; ++x) Y(y0 + y, x0 + x) = (uint8_t)out[y][x];
  }
  void pred16(const MB& m, int mx, int my) {
    const int x0 = mx * 16, y0 = my * 16;
    const bool left = mx > 0, top = my > 0;
    int T[16], L[16];
    for (int i = 0; i < 16; ++i) T[i] = top ? Y(y0 - 1, x0 + i) : 0, L[i] = left ? Y(y0 + i, x0 - 1) : 0;
    if (m.i16 == 0) {
      for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) Y(y0 + y, x0 + x) = (uint8_t)T[x];
    } else if (m.i16 == 1) {
      for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) Y(y0 + y, x0 + x) = (uint8_t)L[y];
    } else if (m.i16 == 2) {
      int st = 0, sl = 0, v;
      for (int i = 0; i < 16; ++i) st += T[i], sl += L[i];
      if (top && left) v = (st + sl + 16) >> 5;
      else if (left) v = (sl + 8) >> 4;
      else if (top) v = (st + 8) >> 4;
      else v = 128;
      for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) Y(y0 + y, x0 + x) = (uint8_t)v;
    } else {
      const int tl = Y(y0 - 1, x0 - 1);
      int Hs = 0, Vs = 0;
      for (int i = 0; i < 8; ++i) Hs += (i + 1) * (T[8 + i] - (6 - i >= 0 ? T[6 - i] : tl)), Vs += (i + 1) * (L[8 + i] - (6 - i >= 0 ? L[6 - i] : tl));
      const int a = 16 * (L[15] + T[15]), b = (5 * Hs + 32) >> 6, c = (5 * Vs + 32) >> 6;
      for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) Y(y0 + y, x0 + x) = (uint8_t)clip1((a + b * (x - 7) + c * (y - 7) + 16) >> 5);
    }
  }
  void pred_chroma(const MB& m, int mx, int my) {
    const bool left = mx > 0, top = my > 0;
    const int x0 = mx * 8, y0 = my * 8, CW = W * 8;
    for (int comp = 0; comp < 2; ++comp) {
      uint8_t* P = C(comp);
      auto px = [&](int y, int x) -> uint8_t& { return P[(size_t)y * CW + x]; };
      int T[8], L[8];
      for (int i = 0; i < 8; ++i) T[i] = top ? px(y0 - 1, x0 + i) : 0, L[i] = left ? px(y0 + i, x0 - 1) : 0;
      const int mode = m.chroma_mode;
      if (mode == 0) {
        for (int by = 0; by < 2; ++by)
          for (int bx = 0; bx < 2; ++bx) {
            int st = 0, sl = 0, v;
            for (int i = 0; i < 4; ++i) st += T[bx * 4 + i], sl += L[by * 4 + i];
            if (bx == by) {
              if (top && left) v = (st + sl + 4) >> 3;
              else if (top) v = (st + 2) >> 2;
              else if (left) v = (sl + 2) >> 2;
              else v = 128;
            } else if (bx == 1) {
              v = top ? (st + 2) >> 2 : (left ? (sl + 2) >> 2 : 128);
            } else {
              v = left ? (sl + 2) >> 2 : (top ? (st + 2) >> 2 : 128);
            }
            for (int y = 0; y < 4; ++y)
              for (int x = 0; x < 4; ++x) px(y0 + by * 4 + y, x0 + bx * 4 + x) = (uint8_t)v;
          }
      } else if (mode == 1) {
        for (int y = 0; y < 8; ++y)
          for (int x = 0; x < 8; ++x) px(y0 + y, x0 + x) = (uint8_t)L[y];
      } else if (mode == 2) {
        for (int y = 0; y < 8; ++y)
          for (int x = 0; x < 8; ++x) px(y0 + y, x0 + x) = (uint8_t)T[x];
      } else {
        const int tl = px(y0 - 1, x0 - 1);
        int Hs = 0, Vs = 0;
        for (int i = 0; i < 4; ++i) Hs += (i + 1) * (T[4 + i] - (2 - i >= 0 ? T[2 - i] : tl)), Vs += (i + 1) * (L[4 + i] - (2 - i >= 0 ? L[2 - i] : tl));
        const int a = 16 * (L[7] + T[7]), b = (34 * Hs + 32) >> 6, c = (34 * Vs + 32) >> 6;
        for (int y = 0; y < 8; ++y)
          for (int x = 0; x < 8; ++x) px(y0 + y, x0 + x) = (uint8_t)clip1((a + b * (x - 3) + c * (y - 3) + 16) >> 5);
      }
    }
  }
  static void idct8_1d(const int* v, int stride, int* o, int ostride) {
    const int d0 = v[0], d1 = v[stride], d2 = v[2 * stride], d3 = v[3 * stride], d4 = v[4 * stride], d5 = v[5 * stride], d6 = v[6 * stride], d7 = v[7 * stride];
    const int a0 = d0 + d4, a4 = d0 - d4, a2 = (d2 >> 1) - d6, a6 = d2 + (d6 >> 1);
    const int b0 = a0 + a6, b2 = a4 + a2, b4 = a4 - a2, b6 = a0 - a6;
    const int a1 = -d3 + d5 - d7 - (d7 >> 1), a3 = d1 + d7 - d3 - (d3 >> 1), a5 = -d1 + d7 + d5 + (d5 >> 1), a7 = d3 + d5 + d1 + (d1 >> 1);
    const int b1 = a1 + (a7 >> 2), b7 = a7 - (a1 >> 2), b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5;
    o[0] = b0 + b7, o[ostride] = b2 + b5, o[2 * ostride] = b4 + b3, o[3 * ostride] = b6 + b1;
    o[4 * ostride] = b6 - b1, o[5 * ostride] = b4 - b3, o[6 * ostride] = b2 - b5, o[7 * ostride] = b0 - b7;
  }
  static void idct8(const int d[8][8], int r[8][8]) {  // 8.5.13
    int t[8][8];
    for (int y = 0; y < 8; ++y) idct8_1d(&d[y][0], 1, &t[y][0], 1);
    for (int x = 0; x < 8; ++x) idct8_1d(&t[0][x], 8, &r[0][x], 8);
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x) r[y][x] = (r[y][x] + 32) >> 6;
  }
  void pred8(const MB& m, int mx, int my, int b8) {  // Intra 8x8 (8.3.2): filtered reference samples, nine modes
    const int bx = b8 & 1, by = b8 >> 1, x0 = mx * 16 + bx * 8, y0 = my * 16 + by * 8;
    const bool left = bx > 0 || mx > 0, top = by > 0 || my > 0;
    bool tr, tl;
    if (b8 == 0) tr = top, tl = mx > 0 && my > 0;
    else if (b8 == 1) tr = my > 0 && mx + 1 < W, tl = top;
    else if (b8 == 2) tr = true, tl = left;
    else tr = false, tl = true;
    int T[16], L[8], c = 0;
    for (int i = 0; i < 16; ++i) T[i] = 0;
    for (int i = 0; i < 8; ++i) L[i] = 0;
    if (top) {
      for (int i = 0; i < 8; ++i) T[i] = Y(y0 - 1, x0 + i);
      for (int i = 8; i < 16; ++i) T[i] = tr ? Y(y0 - 1, x0 + i) : T[7];
    }
    if (left)
      for (int j = 0; j < 8; ++j) L[j] = Y(y0 + j, x0 - 1);
    if (tl) c = Y(y0 - 1, x0 - 1);
    int FT[16], FL[8], fc = 0;
    for (int i = 0; i < 16; ++i) FT[i] = 0;
    for (int i = 0; i < 8; ++i) FL[i] = 0;
    if (top) {
      FT[0] = tl ? (c + 2 * T[0] + T[1] + 2) >> 2 : (3 * T[0] + T[1] + 2) >> 2;
      for (int x = 1; x < 15; ++x) FT[x] = (T[x - 1] + 2 * T[x] + T[x + 1] + 2) >> 2;
      FT[15] = (T[14] + 3 * T[15] + 2) >> 2;
    }
    if (tl) fc = (top && left) ? (T[0] + 2 * c + L[0] + 2) >> 2 : (top ? (3 * c + T[0] + 2) >> 2 : (left ? (3 * c + L[0] + 2) >> 2 : c));
    if (left) {
      FL[0] = tl ? (c + 2 * L[0] + L[1] + 2) >> 2 : (3 * L[0] + L[1] + 2) >> 2;
      for (int y = 1; y < 7; ++y) FL[y] = (L[y - 1] + 2 * L[y] + L[y + 1] + 2) >> 2;
      FL[7] = (L[6] + 3 * L[7] + 2) >> 2;
    }
    auto P = [&](int x, int y) -> int { return y < 0 ? (x < 0 ? fc : FT[x]) : FL[y]; };
    const int mode = m.modes[b8 * 4];
    int dc = 128;
    if (mode == 2) {
      int st = 0, sl = 0;
      for (int i = 0; i < 8; ++i) st += FT[i], sl += FL[i];
      dc = (top && left) ? (st + sl + 8) >> 4 : (left ? (sl + 4) >> 3 : (top ? (st + 4) >> 3 : 128));
    }
    for (int y = 0; y < 8; ++y)
      for (int x = 0; x < 8; ++x) {
        int v;
        switch (mode) {
          case 0: v = P(x, -1); break;
          case 1: v = P(-1, y); break;
          case 2: v = dc; break;
          case 3: v = (x == 7 && y == 7) ? (P(14, -1) + 3 * P(15, -1) + 2) >> 2 : (P(x + y, -1) + 2 * P(x + y + 1, -1) + P(x + y + 2, -1) + 2) >> 2; break;
          case 4:
            if (x > y) v = (P(x - y - 2, -1) + 2 * P(x - y - 1, -1) + P(x - y, -1) + 2) >> 2;
            else if (x < y) v = (P(-1, y - x - 2) + 2 * P(-1, y - x - 1) + P(-1, y - x) + 2) >> 2;
            else v = (P(0, -1) + 2 * P(-1, -1) + P(-1, 0) + 2) >> 2;
            break;
          case 5: {
            const int z = 2 * x - y;
            if (z >= 0 && z % 2 == 0) v = (P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 1) >> 1;
            else if (z >= 0) v = (P(x - (y >> 1) - 2, -1) + 2 * P(x - (y >> 1) - 1, -1) + P(x - (y >> 1), -1) + 2) >> 2;
            else if (z == -1) v = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
            else v = (P(-1, y - 2 * x - 1) + 2 * P(-1, y - 2 * x - 2) + P(-1, y - 2 * x - 3) + 2) >> 2;
            break;
          }
          case 6: {
            const int z = 2 * y - x;
            if (z >= 0 && z % 2 == 0) v = (P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 1) >> 1;
            else if (z >= 0) v = (P(-1, y - (x >> 1) - 2) + 2 * P(-1, y - (x >> 1) - 1) + P(-1, y - (x >> 1)) + 2) >> 2;
            else if (z == -1) v = (P(-1, 0) + 2 * P(-1, -1) + P(0, -1) + 2) >> 2;
            else v = (P(x - 2 * y - 1, -1) + 2 * P(x - 2 * y - 2, -1) + P(x - 2 * y - 3, -1) + 2) >> 2;
            break;
          }
          case 7:
            v = (y % 2 == 0) ? (P(x + (y >> 1), -1) + P(x + (y >> 1) + 1, -1) + 1) >> 1
                             : (P(x + (y >> 1), -1) + 2 * P(x + (y >> 1) + 1, -1) + P(x + (y >> 1) + 2, -1) + 2) >> 2;
            break;
          default: {
            const int z = x + 2 * y;
            if (z > 13) v = P(-1, 7);
            else if (z == 13) v = (P(-1, 6) + 3 * P(-1, 7) + 2) >> 2;
            else if (z % 2 == 0) v = (P(-1, y + (x >> 1)) + P(-1, y + (x >> 1) + 1) + 1) >> 1;
            else v = (P(-1, y + (x >> 1)) + 2 * P(-1, y + (x >> 1) + 1) + P(-1, y + (x >> 1) + 2) + 2) >> 2;
          }
        }
        Y(y0 + y, x0 + x) = (uint8_t)v;
      }
  }
  static void idct4(const int d[4][4], int r[4][4]) {
    int f[4][4];
    for (int i = 0; i < 4; ++i) {
      const int e0 = d[i][0] + d[i][2], e1 = d[i][0] - d[i][2], e2 = (d[i][1] >> 1) - d[i][3], e3 = d[i][1] + (d[i][3] >> 1);
      f[i][0] = e0 + e3, f[i][1] = e1 + e2, f[i][2] = e1 - e2, f[i][3] = e0 - e3;
    }
    for (int j = 0; j < 4; ++j) {
      const int g0 = f[0][j] + f[2][j], g1 = f[0][j] - f[2][j], g2 = (f[1][j] >> 1) - f[3][j], g3 = f[1][j] + (f[3][j] >> 1);
      r[0][j] = (g0 + g3 + 32) >> 6, r[1][j] = (g1 + g2 + 32) >> 6, r[2][j] = (g1 - g2 + 32) >> 6, r[3][j] = (g0 - g3 + 32) >> 6;
    }
  }

  // ---- syntax elements: CABAC (9.3) / CAVLC (9.2, 7.3.5)
  bool cabac() const { return s->cabac != 0; }
  int read_vlc(const uint8_t* lens, const uint8_t* codes, int n, int maxlen) {
    int code = 0;
    for (int ln = 1; ln <= maxlen; ++ln) {
      code = (code << 1) | bits.u1();
      for (int k = 0; k < n; ++k)
        if (lens[k] == ln && codes[k] == code) return k;
    }
    throw Desync{"no CAVLC codeword matches"};
  }
  void intra_mb_type(int base, MB& m) {  // the suffix of mb_type for an intra macroblock in a P (17) / B (32) slice
    if (cab.decision(base) == 0) {
      m.typ = T_I4;
      return;
    }
    H264_CHECK(!cab.terminate(), "I_PCM macroblocks are not implemented");
    m.typ = T_I16;
    const int ac = cab.decision(base + 1);
    int chroma = 0;
    if (cab.decision(base + 2)) chroma = 1 + cab.decision(base + 2);
    int pm = 2 * cab.decision(base + 3);
    pm += cab.decision(base + 3);
    m.i16 = (uint8_t)pm, m.cbp_luma = ac ? 15 : 0, m.cbp_chroma = (uint8_t)chroma;
  }
  void sub_mb_types(int shape[4], int pred[4]) {  // shape -1 = direct
    for (int q = 0; q < 4; ++q) {
      if (!cabac()) {
        const int st = bits.ue();
        H264_CHECK(st < 4, "sub_mb_type out of range");
        shape[q] = st, pred[q] = 0;
      } else if (stype == 0) {
        int st;
        if (cab.decision(21)) st = 0;
        else if (cab.decision(22) == 0) st = 1;
        else st = cab.decision(23) == 0 ? 3 : 2;
        shape[q] = st, pred[q] = 0;
      } else {
        if (cab.decision(36) == 0) {
          shape[q] = -1, pred[q] = -1;
          continue;
        }
        int st;
        if (cab.decision(37) == 0) {
          st = 1 + cab.decision(39);
        } else {
          st = 3;
          bool done_ = false;
          if (cab.decision(38)) {
            if (cab.decision(39)) {
              st = 11 + cab.decision(39);
              done_ = true;
            } else {
              st += 4;
            }
          }
          if (!done_) {
            st += 2 * cab.decision(39);
            st += cab.decision(39);
          }
        }
        shape[q] = B_SUB[st][0], pred[q] = B_SUB[st][1];
      }
    }
  }
  int ref_idx(int l, int x4, int y4) {
    if (!cabac()) return s->nref[l] == 2 ? 1 - bits.u1() : bits.ue();
    int ctx = 0;
    if (x4 > 0 && REF(l, y4, x4 - 1) > 0 && !direct[(size_t)y4 * W4 + x4 - 1]) ctx += 1;
    if (y4 > 0 && REF(l, y4 - 1, x4) > 0 && !direct[(size_t)(y4 - 1) * W4 + x4]) ctx += 2;
    int v = 0;
    while (cab.decision(54 + ctx)) {
      ++v;
      ctx = (ctx >> 2) + 4;
      H264_CHECK(v < 32, "ref_idx runaway");
    }
    return v;
  }
  int read_mvd(int l, int comp, int x4, int y4) {
    if (!cabac()) return bits.se();
    int sum = 0;
    if (x4 > 0) sum += MVD(l, y4, x4 - 1)[comp];
    if (y4 > 0) sum += MVD(l, y4 - 1, x4)[comp];
    const int base = comp == 0 ? 40 : 47;
    if (!cab.decision(base + (sum < 3 ? 0 : (sum > 32 ? 2 : 1)))) return 0;
    int v = 1, c = base + 3;
    while (v < 9 && cab.decision(c)) {
      if (v < 4) ++c;
      ++v;
    }
    if (v >= 9) {
      int k = 3;
      while (cab.bypass()) {
        v += 1 << k;
        ++k;
        H264_CHECK(k < 24, "mvd runaway");
      }
      while (k) {
        --k;
        v += cab.bypass() << k;
      }
    }
    return cab.bypass() ? -v : v;
  }
  int t8_flag(const MB* A, const MB* B) {
    H264_CHECK(cabac(), "the 8x8 transform with CAVLC entropy coding is not implemented");
    return cab.decision(399 + ((A && A->t8) ? 1 : 0) + ((B && B->t8) ? 1 : 0));
  }
  void residual_block8(int* coef) {  // 64 levels in scan order (ctxBlockCat 5; coded_block_flag inferred 1)
    for (int i = 0; i < 64; ++i) coef[i] = 0;
    int sig[64], ns = 0;
    bool last_found = false;
    for (int i = 0; i < 63; ++i)
      if (cab.decision(402 + SIG8[i])) {
        sig[ns++] = i;
        if (cab.decision(417 + LAST8[i])) {
          last_found = true;
          break;
        }
      }
    if (!last_found) sig[ns++] = 63;
    int eq1 = 0, gt1 = 0;
    for (int k = ns - 1; k >= 0; --k) {
      const int inc = gt1 ? 0 : std::min(4, 1 + eq1);
      int v = 0;
      if (cab.decision(426 + inc)) {
        const int inc2 = 5 + std::min(4, gt1);
        v = 1;
        while (v < 14 && cab.decision(426 + inc2)) ++v;
        if (v == 14) {
          int kk = 0;
          while (cab.bypass()) {
            v += 1 << kk;
            ++kk;
            H264_CHECK(kk < 24, "coefficient runaway");
          }
          while (kk) {
            --kk;
            v += cab.bypass() << kk;
          }
        }
      }
      if (v == 0) ++eq1; else ++gt1;
      coef[sig[k]] = cab.bypass() ? -(v + 1) : v + 1;
    }
  }
  int i4_mode() {  // -1: use the predicted mode
    if (!cabac()) return bits.u1() ? -1 : bits.u(3);
    if (cab.decision(68)) return -1;
    int r = cab.decision(69);
    r |= cab.decision(69) << 1;
    r |= cab.decision(69) << 2;
    return r;
  }
  int read_chroma_mode(const MB* A, const MB* B) {
    if (!cabac()) {
      const int v = bits.ue();
      H264_CHECK(v < 4, "intra_chroma_pred_mode out of range");
      return v;
    }
    const int inc = ((A && A->chroma_mode != 0) ? 1 : 0) + ((B && B->chroma_mode != 0) ? 1 : 0);
    int cm = 0;
    if (cab.decision(64 + inc)) {
      cm = 1;
      if (cab.decision(64 + 3)) {
        cm = 2;
        if (cab.decision(64 + 3)) cm = 3;
      }
    }
    return cm;
  }
  void read_cbp(MB& m, const MB* A, const MB* B) {
    if (!cabac()) {
      const int v = bits.ue();
      H264_CHECK(v < 48, "coded_block_pattern out of range");
      const int cbp = (m.intra ? CBP_INTRA : CBP_INTER)[v];
      m.cbp_luma = cbp & 15, m.cbp_chroma = (uint8_t)(cbp >> 4);
      return;
    }
    int cbp = 0;
    for (int b8 = 0; b8 < 4; ++b8) {
      const int x8 = b8 & 1, y8 = b8 >> 1;
      auto cond = [&](int dx, int dy) -> int {
        const int x = x8 + dx, y = y8 + dy;
        if (x >= 0 && x < 2 && y >= 0 && y < 2) return ((cbp >> (y * 2 + x)) & 1) ? 0 : 1;
        const MB* n = dx ? A : B;
        if (!n) return 0;
        return ((n->cbp_luma >> ((((y % 2) + 2) % 2) * 2 + (((x % 2) + 2) % 2))) & 1) ? 0 : 1;
      };
      if (cab.decision(73 + cond(-1, 0) + 2 * cond(0, -1))) cbp |= 1 << b8;
    }
    m.cbp_luma = (uint8_t)cbp;
    int ca = (A && A->cbp_chroma != 0) ? 1 : 0, cb = (B && B->cbp_chroma != 0) ? 1 : 0;
    if (cab.decision(77 + ca + 2 * cb)) {
      ca = (A && A->cbp_chroma == 2) ? 1 : 0, cb = (B && B->cbp_chroma == 2) ? 1 : 0;
      m.cbp_chroma = (uint8_t)(1 + cab.decision(77 + 4 + ca + 2 * cb));
    }
  }
  void read_qp_delta(MB& m, bool coded) {
    if (coded) {
      int dqp;
      if (!cabac()) {
        dqp = bits.se();
        H264_CHECK(dqp >= -26 && dqp <= 25, "mb_qp_delta out of range");
      } else {
        int k = 0;
        if (cab.decision(60 + prev_qp_delta_nz)) {
          k = 1;
          if (cab.decision(60 + 2)) {
            k = 2;
            while (cab.decision(60 + 3)) {
              ++k;
              H264_CHECK(k < 120, "mb_qp_delta runaway");
            }
          }
        }
        dqp = (k & 1) ? (k + 1) / 2 : -(k / 2);
      }
      qp = (qp + dqp + 52) % 52;
      m.qp_delta_nz = dqp ? 1 : 0;
    }
    prev_qp_delta_nz = m.qp_delta_nz;
    m.qp = (int8_t)qp;
  }

  // residual block -> coef[n_coef] in scan order; returns coded_block_flag
  int residual_block(MB& m, const MB* A, const MB* B, int cat, int n_coef, int bx, int by, int comp, int* coef) {
    for (int i = 0; i < n_coef; ++i) coef[i] = 0;
    if (!cabac()) return residual_cavlc(cat, n_coef, bx, by, comp, coef);
    auto cbf_of = [&](const MB* n, int blk) -> int {  // -1 = "not available inside an available macroblock"
      if (cat == 0) return n->typ == T_I16 ? n->cbf_dc : -1;
      if (cat == 1 || cat == 2) return ((n->cbp_luma >> ((BLK_Y[blk] >> 1) * 2 + (BLK_X[blk] >> 1))) & 1) ? n->cbf_luma[blk] : -1;
      if (cat == 3) return n->cbp_chroma ? n->cbf_cdc[comp] : -1;
      return n->cbp_chroma == 2 ? n->cbf_cac[comp][blk] : -1;
    };
    const int size = (cat == 1 || cat == 2) ? 4 : (cat == 4 ? 2 : 1);
    if (cat == 0 || cat == 3) bx = by = 0;
    auto flag = [&](int dx, int dy) -> int {
      const int x = bx + dx, y = by + dy;
      const MB* n;
      int xx, yy;
      if (x >= 0 && x < size && y >= 0 && y < size) {
        n = &m, xx = x, yy = y;
      } else {
        n = dx ? A : B;
        if (!n) return m.intra ? 1 : 0;
        xx = ((x % size) + size) % size, yy = ((y % size) + size) % size;
      }
      const int blk = (cat == 1 || cat == 2) ? XY_BLK[yy][xx] : (cat == 4 ? yy * 2 + xx : 0);
      const int v = cbf_of(n, blk);
      return v < 0 ? 0 : v;
    };
    const int fa = flag(-1, 0), fb = flag(0, -1);
    if (!cab.decision(85 + CAT_CBF[cat] + fa + 2 * fb)) return 0;
    int sig[16], ns = 0;
    bool last_found = false;
    for (int i = 0; i < n_coef - 1; ++i) {
      const int inc = cat == 3 ? std::min(i, 2) : i;
      if (cab.decision(105 + CAT_SIG[cat] + inc)) {
        sig[ns++] = i;
        if (cab.decision(166 + CAT_SIG[cat] + inc)) {
          last_found = true;
          break;
        }
      }
    }
    if (!last_found) sig[ns++] = n_coef - 1;
    int eq1 = 0, gt1 = 0;
    for (int k = ns - 1; k >= 0; --k) {
      const int ctx0 = 227 + CAT_ABS[cat];
      const int inc = gt1 ? 0 : std::min(4, 1 + eq1);
      int v = 0;
      if (cab.decision(ctx0 + inc)) {
        const int inc2 = 5 + std::min(4 - (cat == 3 ? 1 : 0), gt1);
        v = 1;
        while (v < 14 && cab.decision(ctx0 + inc2)) ++v;
        if (v == 14) {
          int kk = 0;
          while (cab.bypass()) {
            v += 1 << kk;
            ++kk;
            H264_CHECK(kk < 24, "coefficient runaway");
          }
          while (kk) {
            --kk;
            v += cab.bypass() << kk;
          }
        }
      }
      if (v == 0) ++eq1; else ++gt1;
      coef[sig[k]] = cab.bypass() ? -(v + 1) : v + 1;
    }
    return 1;
  }
  int residual_cavlc(int cat, int n_coef, int bx, int by, int comp, int* coef) {
    const int mx = cur_mx, my = cur_my;
    int k;
    if (cat == 3) {
      k = read_vlc(CDC_LEN, CDC_BITS, 20, 8);
    } else {
      const std::vector<int32_t>& arr = cat == 4 ? tcc[comp] : tc;
      const int aw = cat == 4 ? W * 2 : W4;
      const int gx = cat == 4 ? mx * 2 + bx : mx * 4 + (cat == 0 ? 0 : bx), gy = cat == 4 ? my * 2 + by : my * 4 + (cat == 0 ? 0 : by);
      const int na = gx > 0 ? arr[(size_t)gy * aw + gx - 1] : -1, nb_ = gy > 0 ? arr[(size_t)(gy - 1) * aw + gx] : -1;
      const int nc = (na >= 0 && nb_ >= 0) ? (na + nb_ + 1) >> 1 : (na >= 0 ? na : (nb_ >= 0 ? nb_ : 0));
      const int t = nc < 2 ? 0 : (nc < 4 ? 1 : (nc < 8 ? 2 : 3));
      k = read_vlc(CT_LEN[t], CT_BITS[t], 68, 16);
    }
    const int total = k >> 2, t1 = k & 3;
    if (cat == 1 || cat == 2) tc[(size_t)(my * 4 + by) * W4 + mx * 4 + bx] = total;
    else if (cat == 4) tcc[comp][(size_t)(my * 2 + by) * (W * 2) + mx * 2 + bx] = total;
    if (total == 0) return 0;
    H264_CHECK(total <= n_coef && t1 <= std::min(total, 3), "coeff_token out of range");
    int levels[16];
    int suffix_len = (total > 10 && t1 < 3) ? 1 : 0;
    for (int i = 0; i < total; ++i) {
      if (i < t1) {
        levels[i] = 1 - 2 * bits.u1();
        continue;
      }
      int prefix = 0;
      while (bits.u1() == 0) {
        ++prefix;
        H264_CHECK(prefix < 32, "level_prefix runaway");
      }
      int code = std::min(15, prefix) << suffix_len;
      if (suffix_len > 0 || prefix >= 14) {
        const int size = (prefix == 14 && suffix_len == 0) ? 4 : (prefix >= 15 ? prefix - 3 : suffix_len);
        if (size) code += bits.u(size);
      }
      if (prefix >= 15 && suffix_len == 0) code += 15;
      if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
      if (i == t1 && t1 < 3) code += 2;
      const int lv = (code % 2 == 0) ? (code + 2) >> 1 : (-code - 1) >> 1;
      levels[i] = lv;
      if (suffix_len == 0) suffix_len = 1;
      if (std::abs(lv) > (3 << (suffix_len - 1)) && suffix_len < 6) ++suffix_len;
    }
    int zeros_left = 0;
    if (total < n_coef) {
      zeros_left = cat == 3 ? read_vlc(CTZ_LEN[total - 1], CTZ_BITS[total - 1], 4, 3) : read_vlc(TZ_LEN[total - 1], TZ_BITS[total - 1], TZ_N[total - 1], 9);
      H264_CHECK(total + zeros_left <= n_coef, "total_zeros out of range");
    }
    int pos = total + zeros_left - 1;
    for (int i = 0; i < total; ++i) {
      coef[pos] = levels[i];
      if (i < total - 1) {
        int run = 0;
        if (zeros_left > 0) {
          const int t = std::min(zeros_left, 7) - 1;
          run = read_vlc(RUN_LEN[t], RUN_BITS[t], RUN_N[t], 11);
        }
        H264_CHECK(run <= zeros_left, "run_before out of range");
        zeros_left -= run;
        pos -= 1 + run;
      }
    }
    return 1;
  }

  // ---- reconstruction of one macroblock's residual (intra prediction interleaved)
  void residual(int mx, int my, MB& m, const MB* A, const MB* B) {
    const int qpy = m.qp, px = mx * 16, py = my * 16, X4 = mx * 4, Y4 = my * 4;
    int lv[16], d[4][4], r[4][4];
    int dc16[4][4];
    bool have_dc = false;
    if (m.typ == T_I16) {
      m.cbf_dc = (uint8_t)residual_block(m, A, B, 0, 16, 0, 0, 0, lv);
      int c[4][4], t[4][4], f[4][4];
      static const int Am[4][4] = {{1, 1, 1, 1}, {1, 1, -1, -1}, {1, -1, -1, 1}, {1, -1, 1, -1}};
      for (int k = 0; k < 16; ++k) c[ZZ_Y[k]][ZZ_X[k]] = lv[k];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          t[i][j] = 0;
          for (int k = 0; k < 4; ++k) t[i][j] += Am[i][k] * c[k][j];
        }
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          f[i][j] = 0;
          for (int k = 0; k < 4; ++k) f[i][j] += t[i][k] * Am[k][j];
        }
      const int ls = level_scale(qpy, 0, 0);
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
          dc16[i][j] = qpy >= 36 ? (f[i][j] * ls) * (1 << (qpy / 6 - 6)) : (f[i][j] * ls + (1 << (5 - qpy / 6))) >> (6 - qpy / 6);
      have_dc = true;
      pred16(m, mx, my);
    }
    if (m.t8) {
      // 8x8 transform: four luma blocks of 64 levels (coded_block_flag inferred), Intra 8x8 prediction interleaved
      int lv8[64], d8[8][8], r8[8][8];
      for (int b8 = 0; b8 < 4; ++b8) {
        const int bx8 = b8 & 1, by8 = b8 >> 1;
        if (m.typ == T_I8) pred8(m, mx, my, b8);
        if (!((m.cbp_luma >> b8) & 1)) continue;
        residual_block8(lv8);
        memset(d8, 0, sizeof(d8));
        // 8x8 zig-zag: anti-diagonals, odd ones from the top right down, even ones from the bottom left up
        int k = 0;
        for (int dg = 0; dg < 15; ++dg) {
          const int lo = std::max(dg - 7, 0), hi = std::min(dg, 7);
          for (int t = 0; t <= hi - lo; ++t, ++k) {
            const int x = (dg % 2) ? hi - t : lo + t, y = dg - x;
            if (lv8[k]) {
              const int ls8 = level_scale8(qpy, y, x);
              d8[y][x] = qpy >= 36 ? (lv8[k] * ls8) * (1 << (qpy / 6 - 6)) : (lv8[k] * ls8 + (1 << (5 - qpy / 6))) >> (6 - qpy / 6);
            }
          }
        }
        idct8(d8, r8);
        for (int y = 0; y < 8; ++y)
          for (int x = 0; x < 8; ++x) {
            uint8_t& p = Y(py + by8 * 8 + y, px + bx8 * 8 + x);
            p = (uint8_t)clip1(p + r8[y][x]);
          }
        for (int q = 0; q < 4; ++q) m.cbf_luma[b8 * 4 + q] = 1;
        for (int y = 0; y < 2; ++y)
          for (int x = 0; x < 2; ++x) nz[(size_t)(Y4 + by8 * 2 + y) * W4 + X4 + bx8 * 2 + x] = 1;
      }
    }
    for (int blk = 0; blk < (m.t8 ? 0 : 16); ++blk) {
      const int bx = BLK_X[blk], by = BLK_Y[blk];
      bool coded = false;
      memset(d, 0, sizeof(d));
      if ((m.cbp_luma >> ((by >> 1) * 2 + (bx >> 1))) & 1) {
        if (m.typ == T_I16) {
          m.cbf_luma[blk] = (uint8_t)residual_block(m, A, B, 1, 15, bx, by, 0, lv + 1);
          lv[0] = 0;
        } else {
          m.cbf_luma[blk] = (uint8_t)residual_block(m, A, B, 2, 16, bx, by, 0, lv);
        }
        coded = m.cbf_luma[blk] == 1;
        if (coded) {
          for (int k = 0; k < 16; ++k)
            if (lv[k]) {
              const int x = ZZ_X[k], y = ZZ_Y[k], lsx = level_scale(qpy, x, y);
              d[y][x] = qpy >= 24 ? (lv[k] * lsx) * (1 << (qpy / 6 - 4)) : (lv[k] * lsx + (1 << (3 - qpy / 6))) >> (4 - qpy / 6);
            }
          nz[(size_t)(Y4 + by) * W4 + X4 + bx] = 1;
        }
      }
      if (m.typ == T_I4) pred4(m, mx, my, blk);
      if (have_dc && dc16[by][bx] != 0) {
        d[0][0] = dc16[by][bx];
        coded = true;
      }
      if (coded) {
        idct4(d, r);
        for (int y = 0; y < 4; ++y)
          for (int x = 0; x < 4; ++x) {
            uint8_t& p = Y(py + by * 4 + y, px + bx * 4 + x);
            p = (uint8_t)clip1(p + r[y][x]);
          }
      }
    }
    if (m.intra) pred_chroma(m, mx, my);
    const int qpcs[2] = {QPC[clip3(0, 51, qpy + s->chroma_qp_offset)], QPC[clip3(0, 51, qpy + s->chroma_qp_offset_cr)]};
    int dcs[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    if (m.cbp_chroma) {
      for (int comp = 0; comp < 2; ++comp) {
        const int qpc = qpcs[comp];
        m.cbf_cdc[comp] = (uint8_t)residual_block(m, A, B, 3, 4, 0, 0, comp, lv);
        const int c00 = lv[0], c01 = lv[1], c10 = lv[2], c11 = lv[3];
        const int f[4] = {c00 + c01 + c10 + c11, c00 - c01 + c10 - c11, c00 + c01 - c10 - c11, c00 - c01 - c10 + c11};
        const int ls = level_scale(qpc, 0, 0);
        for (int i = 0; i < 4; ++i) dcs[comp][i] = ((f[i] * ls) * (1 << (qpc / 6))) >> 5;
      }
    }
    int acs[2][4][16];
    bool have_ac = false;
    if (m.cbp_chroma == 2) {
      have_ac = true;
      for (int comp = 0; comp < 2; ++comp)
        for (int blk = 0; blk < 4; ++blk) {
          m.cbf_cac[comp][blk] = (uint8_t)residual_block(m, A, B, 4, 15, blk & 1, blk >> 1, comp, acs[comp][blk] + 1);
          acs[comp][blk][0] = 0;
        }
    }
    const int CW = W * 8;
    for (int comp = 0; comp < 2; ++comp)
      for (int blk = 0; blk < 4; ++blk) {
        const int bx = blk & 1, by = blk >> 1, qpc = qpcs[comp];
        memset(d, 0, sizeof(d));
        bool any = false;
        if (have_ac)
          for (int k = 0; k < 16; ++k) {
            const int v = acs[comp][blk][k];
            if (v) {
              const int x = ZZ_X[k], y = ZZ_Y[k], lsx = level_scale(qpc, x, y);
              d[y][x] = qpc >= 24 ? (v * lsx) * (1 << (qpc / 6 - 4)) : (v * lsx + (1 << (3 - qpc / 6))) >> (4 - qpc / 6);
              any = true;
            }
          }
        d[0][0] = dcs[comp][blk];
        if (any || d[0][0]) {
          idct4(d, r);
          uint8_t* P = C(comp);
          for (int y = 0; y < 4; ++y)
            for (int x = 0; x < 4; ++x) {
              uint8_t& p = P[(size_t)(my * 8 + by * 4 + y) * CW + mx * 8 + bx * 4 + x];
              p = (uint8_t)clip1(p + r[y][x]);
            }
        }
      }
  }

  // ---- macroblocks
  void mark_done(int mx, int my, int sx, int sy, int w, int h, uint8_t v) {
    for (int y = sy; y < sy + h; ++y)
      for (int x = sx; x < sx + w; ++x) done[(size_t)(my * 4 + y) * W4 + mx * 4 + x] = v;
  }
  int direct_parts(Part* parts, int n, int q_mask) {  // the MC partitions of direct-predicted quadrants
    if (q_mask == 15 && n == 0) {
      // a wholly direct macroblock whose sixteen blocks ended up with the same motion (static background: the usual case) is
      // predicted as ONE 16 x 16 partition -- the same samples as sixteen or four separate blocks, a quarter of the calls
      const int X4 = cur_dx4, Y4 = cur_dy4;
      bool same = true;
      for (int l = 0; l < 2 && same; ++l) {
        const int r = REF(l, Y4, X4);
        const int16_t* m0 = MV(l, Y4, X4);
        for (int y = 0; y < 4 && same; ++y)
          for (int x = 0; x < 4; ++x)
            if (REF(l, Y4 + y, X4 + x) != r || MV(l, Y4 + y, X4 + x)[0] != m0[0] || MV(l, Y4 + y, X4 + x)[1] != m0[1]) {
              same = false;
              break;
            }
      }
      if (same) {
        parts[n++] = {0, 0, 4, 4, 0, 0, 0, 0};
        return n;
      }
    }
    for (int q = 0; q < 4; ++q) {
      if (!((q_mask >> q) & 1)) continue;
      const int qx = (q & 1) * 2, qy = (q >> 1) * 2;
      if (s->direct_8x8_inference) {
        parts[n++] = {qx, qy, 2, 2, 0, 0, 0, q};
      } else {
        for (int k = 0; k < 4; ++k) parts[n++] = {qx + (k & 1), qy + (k >> 1), 1, 1, 0, 0, 0, q};
      }
    }
    return n;
  }
  void skip_mb(int addr, int mx, int my) {
    MB& m = mbs[addr];
    m = MB();
    const int X4 = mx * 4, Y4 = my * 4;
    m.skip = 1, m.typ = T_INTER, m.qp = (int8_t)qp;
    prev_qp_delta_nz = 0;
    ++stats[2];
    Part parts[16];
    int np = 0;
    if (stype == 0) {
      Nb a = nb(0, X4 - 1, Y4), b = nb(0, X4, Y4 - 1);
      int mvx = 0, mvy = 0;
      if (!(!a.avail || !b.avail || (a.ref == 0 && a.mvx == 0 && a.mvy == 0) || (b.ref == 0 && b.mvx == 0 && b.mvy == 0))) mvp(0, X4, Y4, 4, 0, 0, 0, mvx, mvy);
      set_motion(0, X4, Y4, 4, 4, 0, mvx, mvy);
      parts[np++] = {0, 0, 4, 4, 0, 0, 0, 0};
    } else {
      m.direct16 = 1;
      const int quads[4] = {0, 1, 2, 3};
      direct_pred(mx, my, quads, 4);
      cur_dx4 = mx * 4, cur_dy4 = my * 4;
      np = direct_parts(parts, 0, 15);
    }
    mark_done(mx, my, 0, 0, 4, 4, 1);
    predict_inter(mx, my, parts, np);
  }
  void intra_tail(int addr, int mx, int my, MB& m, const MB* A, const MB* B) {
    const int X4 = mx * 4, Y4 = my * 4;
    m.intra = 1;
    for (int y = 0; y < 4; ++y)
      for (int x = 0; x < 4; ++x) cur->intra4[(size_t)(Y4 + y) * W4 + X4 + x] = 1;
    if (m.typ == T_I4 && s->transform_8x8_mode && t8_flag(A, B)) m.typ = T_I8, m.t8 = 1;
    ++stats[m.typ == T_I4 ? 0 : (m.typ == T_I8 ? 5 : 1)];
    mark_done(mx, my, 0, 0, 4, 4, 1);
    if (m.typ == T_I4 || m.typ == T_I8) {
      // Intra 8x8: four blocks whose neighbours are the 4x4 blocks left of / above their first 4x4 block; a mode is kept per 4x4 slot
      for (int blk = 0; blk < 16; blk += (m.typ == T_I4 ? 1 : 4)) {
        const int bx = BLK_X[blk], by = BLK_Y[blk];
        auto nmode = [&](int dx, int dy) -> int {  // -1: not available
          const int x = bx + dx, y = by + dy;
          const MB* n;
          if (x >= 0 && x < 4 && y >= 0 && y < 4) n = &m;
          else n = mb(mx + (x < 0 ? -1 : (x > 3 ? 1 : 0)), my + (y < 0 ? -1 : (y > 3 ? 1 : 0)));
          if (!n) return -1;
          if (n->typ != T_I4 && n->typ != T_I8) return 2;
          return n->modes[XY_BLK[(y + 4) % 4][(x + 4) % 4]];
        };
        const int ma = nmode(-1, 0), mb_ = nmode(0, -1);
        const int pred = (ma < 0 || mb_ < 0) ? 2 : std::min(ma, mb_);
        const int rem = i4_mode();
        const uint8_t mode = (uint8_t)(rem < 0 ? pred : (rem < pred ? rem : rem + 1));
        if (m.typ == T_I4) m.modes[blk] = mode;
        else m.modes[blk] = m.modes[blk + 1] = m.modes[blk + 2] = m.modes[blk + 3] = mode;
      }
    }
    m.chroma_mode = (uint8_t)read_chroma_mode(A, B);
    if (m.typ == T_I4 || m.typ == T_I8) read_cbp(m, A, B);
    read_qp_delta(m, m.typ == T_I16 || m.cbp_luma || m.cbp_chroma);
    residual(mx, my, m, A, B);
  }
  // inter: kind 0 = (shape, preds), 1 = 8x8, 2 = direct 16x16
  void inter_tail(int addr, int mx, int my, MB& m, const MB* A, const MB* B, int kind, int shape, int p0, int p1) {
    const int X4 = mx * 4, Y4 = my * 4;
    m.typ = T_INTER;
    ++stats[3];
    Part plist[16], parts[20];
    int npl = 0, np = 0;
    bool no_sub8 = kind == 2 ? s->direct_8x8_inference != 0 : true;  // noSubMbPartSizeLessThan8x8Flag (7.3.5)
    if (kind == 2) {
      const int quads[4] = {0, 1, 2, 3};
      direct_pred(mx, my, quads, 4);
      cur_dx4 = mx * 4, cur_dy4 = my * 4;
      np = direct_parts(parts, 0, 15);
      mark_done(mx, my, 0, 0, 4, 4, 1);
    } else {
      struct Group { int g, x, y, w, h, pred; } groups[4];
      int ng = 0, dmask = 0;
      static const int SHP[4][4][4] = {{{0, 0, 2, 2}}, {{0, 0, 2, 1}, {0, 1, 2, 1}}, {{0, 0, 1, 2}, {1, 0, 1, 2}}, {{0, 0, 1, 1}, {1, 0, 1, 1}, {0, 1, 1, 1}, {1, 1, 1, 1}}};
      static const int SHN[4] = {1, 2, 2, 4};
      if (kind == 1) {
        int sshape[4], spred[4];
        sub_mb_types(sshape, spred);
        for (int q = 0; q < 4; ++q) no_sub8 = no_sub8 && (sshape[q] == 0 || (sshape[q] < 0 && s->direct_8x8_inference));
        int dq[4], ndq = 0;
        for (int q = 0; q < 4; ++q)
          if (sshape[q] < 0) dq[ndq++] = q, dmask |= 1 << q;
        if (ndq) direct_pred(mx, my, dq, ndq);
        for (int q = 0; q < 4; ++q) {
          if (sshape[q] < 0) continue;
          const int qx = (q & 1) * 2, qy = (q >> 1) * 2;
          for (int k = 0; k < SHN[sshape[q]]; ++k) {
            const int* sp = SHP[sshape[q]][k];
            plist[npl++] = {qx + sp[0], qy + sp[1], sp[2], sp[3], spred[q], 0, 0, q};
          }
          groups[ng++] = {q, qx, qy, 2, 2, spred[q]};
        }
      } else {
        static const int MBP[3][2][4] = {{{0, 0, 4, 4}, {0, 0, 0, 0}}, {{0, 0, 4, 2}, {0, 2, 4, 2}}, {{0, 0, 2, 4}, {2, 0, 2, 4}}};
        const int n = shape == 0 ? 1 : 2;
        for (int i = 0; i < n; ++i) {
          const int* sp = MBP[shape][i];
          const int pr = i ? p1 : p0;
          plist[npl++] = {sp[0], sp[1], sp[2], sp[3], pr, shape, i, i};
          groups[ng++] = {i, sp[0], sp[1], sp[2], sp[3], pr};
        }
      }
      int refs[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
      for (int l = 0; l < 2; ++l)
        for (int gi = 0; gi < ng; ++gi) {
          const Group& g = groups[gi];
          if (g.pred == l || g.pred == 2) {
            const int rf = (s->nref[l] > 1 && !m.ref0) ? ref_idx(l, X4 + g.x, Y4 + g.y) : 0;
            H264_CHECK(rf < s->nref[l], "ref_idx beyond the list");
            refs[l][g.g] = rf;
            for (int y = g.y; y < g.y + g.h; ++y)
              for (int x = g.x; x < g.x + g.w; ++x) REF(l, Y4 + y, X4 + x) = (int8_t)rf;
          }
        }
      for (int l = 0; l < 2; ++l) {
        mark_done(mx, my, 0, 0, 4, 4, 0);
        int qdone = -1;
        for (int k = 0; k < npl; ++k) {
          const Part& p = plist[k];
          if (kind == 1)
            for (int q = 0; q < p.g; ++q)
              if (q > qdone) {
                mark_done(mx, my, (q & 1) * 2, (q >> 1) * 2, 2, 2, 1);
                qdone = q;
              }
          if (p.pred == l || p.pred == 2) {
            const int rf = refs[l][p.g];
            const int dx = read_mvd(l, 0, X4 + p.sx, Y4 + p.sy), dy = read_mvd(l, 1, X4 + p.sx, Y4 + p.sy);
            int px, py;
            mvp(l, X4 + p.sx, Y4 + p.sy, p.w, rf, p.shape, p.pi, px, py);
            set_motion(l, X4 + p.sx, Y4 + p.sy, p.w, p.h, rf, px + dx, py + dy, dx, dy);
          }
          mark_done(mx, my, p.sx, p.sy, p.w, p.h, 1);
        }
        mark_done(mx, my, 0, 0, 4, 4, 1);
      }
      for (int k = 0; k < npl; ++k) parts[np++] = plist[k];
      if (kind == 1) {
        cur_dx4 = mx * 4, cur_dy4 = my * 4;
        np = direct_parts(parts, np, dmask);
      }
    }
    predict_inter(mx, my, parts, np);
    read_cbp(m, A, B);
    if (s->transform_8x8_mode && m.cbp_luma && no_sub8 && t8_flag(A, B)) {
      m.t8 = 1;
      ++stats[6];
    }
    read_qp_delta(m, m.cbp_luma || m.cbp_chroma);
    residual(mx, my, m, A, B);
  }
  void macroblock_cabac(int addr, int mx, int my) {
    MB* A = mb(mx - 1, my);
    MB* B = mb(mx, my - 1);
    MB& m = mbs[addr];
    m = MB();
    if (stype != 2) {
      const int ctx = (stype == 0 ? 11 : 24) + ((A && !A->skip) ? 1 : 0) + ((B && !B->skip) ? 1 : 0);
      if (cab.decision(ctx)) {
        skip_mb(addr, mx, my);
        return;
      }
    }
    int kind = -1, shape = 0, p0 = 0, p1 = -1;
    if (stype == 2) {
      const int inc = ((A && A->typ != T_I4 && A->typ != T_I8) ? 1 : 0) + ((B && B->typ != T_I4 && B->typ != T_I8) ? 1 : 0);
      if (cab.decision(3 + inc) == 0) {
        m.typ = T_I4;
      } else {
        H264_CHECK(!cab.terminate(), "I_PCM macroblocks are not implemented");
        m.typ = T_I16;
        const int ac = cab.decision(3 + 3);
        int chroma = 0;
        if (cab.decision(3 + 4)) chroma = 1 + cab.decision(3 + 5);
        int pm = 2 * cab.decision(3 + 6);
        pm += cab.decision(3 + 7);
        m.i16 = (uint8_t)pm, m.cbp_luma = ac ? 15 : 0, m.cbp_chroma = (uint8_t)chroma;
      }
    } else if (stype == 0) {
      if (cab.decision(14) == 0) {
        if (cab.decision(15) == 0) {
          if (cab.decision(16)) kind = 1; else kind = 0, shape = 0;
        } else {
          kind = 0, shape = cab.decision(17) ? 1 : 2, p1 = 0;
        }
      } else {
        intra_mb_type(17, m);
      }
    } else {
      const int inc = ((A && !A->direct16) ? 1 : 0) + ((B && !B->direct16) ? 1 : 0);
      int t;
      if (cab.decision(27 + inc) == 0) {
        t = 0;
      } else if (cab.decision(27 + 3) == 0) {
        t = 1 + cab.decision(27 + 5);
      } else {
        int b = cab.decision(27 + 4) << 3;
        b |= cab.decision(27 + 5) << 2;
        b |= cab.decision(27 + 5) << 1;
        b |= cab.decision(27 + 5);
        if (b < 8) t = b + 3;
        else if (b == 13) t = 23;
        else if (b == 14) t = 11;
        else if (b == 15) t = 22;
        else t = ((b << 1) | cab.decision(27 + 5)) - 4;
      }
      if (t == 23) intra_mb_type(32, m);
      else if (t == 22) kind = 1;
      else if (t == 0) kind = 2, m.direct16 = 1;
      else kind = 0, shape = B_MB[t][0], p0 = B_MB[t][1], p1 = B_MB[t][2];
    }
    if (kind < 0) intra_tail(addr, mx, my, m, A, B);
    else inter_tail(addr, mx, my, m, A, B, kind, shape, p0, p1);
  }
  void macroblock_cavlc(int addr, int mx, int my) {
    MB* A = mb(mx - 1, my);
    MB* B = mb(mx, my - 1);
    MB& m = mbs[addr];
    m = MB();
    cur_mx = mx, cur_my = my;
    int t = bits.ue();
    int kind = -1, shape = 0, p1 = -1;
    if (stype == 0) {
      if (t < 5) {
        if (t >= 3) kind = 1, m.ref0 = t == 4;
        else kind = 0, shape = t, p1 = t ? 0 : -1;
      } else {
        t -= 5;
      }
    }
    if (kind < 0) {
      if (t == 0) {
        m.typ = T_I4;
      } else {
        H264_CHECK(t < 25, "mb_type out of range or I_PCM");
        m.typ = T_I16;
        m.i16 = (uint8_t)((t - 1) % 4), m.cbp_chroma = (uint8_t)(((t - 1) / 4) % 3), m.cbp_luma = t >= 13 ? 15 : 0;
      }
      intra_tail(addr, mx, my, m, A, B);
    } else {
      inter_tail(addr, mx, my, m, A, B, kind, shape, 0, p1);
    }
  }

  // ---- edge filter (8.7)
  int bs(int py4, int px4, int qy4, int qx4, bool mb_edge) {
    const size_t pi = (size_t)py4 * W4 + px4, qi = (size_t)qy4 * W4 + qx4;
    if (cur->intra4[pi] || cur->intra4[qi]) return mb_edge ? 4 : 3;
    if (nz[pi] || nz[qi]) return 2;
    int32_t pr[2], qr[2];
    int pm[2][2], qm[2][2], np_ = 0, nq = 0;
    for (int l = 0; l < 2; ++l) {
      if (REF(l, py4, px4) >= 0) pr[np_] = REFID(l, py4, px4), pm[np_][0] = MV(l, py4, px4)[0], pm[np_][1] = MV(l, py4, px4)[1], ++np_;
      if (REF(l, qy4, qx4) >= 0) qr[nq] = REFID(l, qy4, qx4), qm[nq][0] = MV(l, qy4, qx4)[0], qm[nq][1] = MV(l, qy4, qx4)[1], ++nq;
    }
    if (np_ != nq) return 1;
    if (np_ == 1) {
      if (pr[0] != qr[0]) return 1;
      return (std::abs(pm[0][0] - qm[0][0]) >= 4 || std::abs(pm[0][1] - qm[0][1]) >= 4) ? 1 : 0;
    }
    if (!((pr[0] == qr[0] && pr[1] == qr[1]) || (pr[0] == qr[1] && pr[1] == qr[0]))) return 1;
    auto far = [&](int a, int b) { return std::abs(pm[a][0] - qm[b][0]) >= 4 || std::abs(pm[a][1] - qm[b][1]) >= 4; };
    if (pr[0] != pr[1]) {
      const int q0 = qr[0] == pr[0] ? 0 : 1;
      return (far(0, q0) || far(1, 1 - q0)) ? 1 : 0;
    }
    return ((far(0, 0) || far(1, 1)) && (far(0, 1) || far(1, 0))) ? 1 : 0;
  }
  static bool filter_line(int* px, int bS, int alpha, int beta, int idx_a, bool luma) {
    const int p3 = px[0], p2 = px[1], p1 = px[2], p0 = px[3], q0 = px[4], q1 = px[5], q2 = px[6], q3 = px[7];
    if (!(std::abs(p0 - q0) < alpha && std::abs(p1 - p0) < beta && std::abs(q1 - q0) < beta)) return false;
    if (bS < 4) {
      const int tc0 = TC0[idx_a][bS - 1];
      int tc, ap = 0, aq = 0;
      if (luma) {
        ap = std::abs(p2 - p0), aq = std::abs(q2 - q0);
        tc = tc0 + (ap < beta ? 1 : 0) + (aq < beta ? 1 : 0);
      } else {
        tc = tc0 + 1;
      }
      const int delta = clip3(-tc, tc, (((q0 - p0) * 4) + (p1 - q1) + 4) >> 3);
      px[3] = clip1(p0 + delta), px[4] = clip1(q0 - delta);
      if (luma) {
        if (ap < beta) px[2] = p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 * 2)) >> 1);
        if (aq < beta) px[5] = q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 * 2)) >> 1);
      }
    } else if (luma) {
      const int ap = std::abs(p2 - p0), aq = std::abs(q2 - q0);
      const bool small = std::abs(p0 - q0) < ((alpha >> 2) + 2);
      if (ap < beta && small) {
        px[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, px[2] = (p2 + p1 + p0 + q0 + 2) >> 2, px[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
      } else {
        px[3] = (2 * p1 + p0 + q1 + 2) >> 2;
      }
      if (aq < beta && small) {
        px[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, px[5] = (p0 + q0 + q1 + q2 + 2) >> 2, px[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
      } else {
        px[4] = (2 * q1 + q0 + p1 + 2) >> 2;
      }
    } else {
      px[3] = (2 * p1 + p0 + q1 + 2) >> 2, px[4] = (2 * q1 + q0 + p1 + 2) >> 2;
    }
    return true;
  }
  void deblock() {
    const int cqo[2] = {s->chroma_qp_offset, s->chroma_qp_offset_cr}, off_a = s->filter_offset_a, off_b = s->filter_offset_b;
    auto qpc = [&](int q, int comp) { return (int)QPC[clip3(0, 51, q + cqo[comp])]; };
    // "flat" macroblocks -- inter, no coded luma block, one motion for all sixteen blocks (static background, skipped macroblocks) --
    // have boundary strength 0 on every inner edge, and on an outer edge whose neighbour is flat with the same motion: most of a
    // typical picture is decided here instead of by thirty-two block-pair comparisons per macroblock
    std::vector<uint8_t> flat((size_t)W * Hh, 0);
    for (int my = 0; my < Hh; ++my)
      for (int mx = 0; mx < W; ++mx) {
        const int X4 = mx * 4, Y4 = my * 4;
        bool f = !mbs[(size_t)my * W + mx].intra;
        for (int l = 0; l < 2 && f; ++l) {
          const int32_t id0 = REFID(l, Y4, X4);
          const int r0 = REF(l, Y4, X4);
          const int16_t* m0 = MV(l, Y4, X4);
          for (int y = 0; y < 4 && f; ++y)
            for (int x = 0; x < 4; ++x)
              if (REF(l, Y4 + y, X4 + x) != r0 || REFID(l, Y4 + y, X4 + x) != id0 || MV(l, Y4 + y, X4 + x)[0] != m0[0] || MV(l, Y4 + y, X4 + x)[1] != m0[1] ||
                  (l == 0 && nz[(size_t)(Y4 + y) * W4 + X4 + x])) {
                f = false;
                break;
              }
        }
        flat[(size_t)my * W + mx] = f;
      }
    auto same_motion = [&](int ax, int ay, int bx, int by) {  // of two flat macroblocks (their first blocks stand for all)
      for (int l = 0; l < 2; ++l)
        if (REF(l, ay * 4, ax * 4) < 0 ? REF(l, by * 4, bx * 4) >= 0
                                       : (REF(l, by * 4, bx * 4) < 0 || REFID(l, ay * 4, ax * 4) != REFID(l, by * 4, bx * 4) ||
                                          MV(l, ay * 4, ax * 4)[0] != MV(l, by * 4, bx * 4)[0] || MV(l, ay * 4, ax * 4)[1] != MV(l, by * 4, bx * 4)[1]))
          return false;
      return true;
    };
    for (int my = 0; my < Hh; ++my)
      for (int mx = 0; mx < W; ++mx) {
        const MB& m = mbs[(size_t)my * W + mx];
        const bool mflat = flat[(size_t)my * W + mx];
        for (int vertical = 1; vertical >= 0; --vertical) {
          const MB* n = vertical ? (mx > 0 ? &mbs[(size_t)my * W + mx - 1] : nullptr) : (my > 0 ? &mbs[(size_t)(my - 1) * W + mx] : nullptr);
          for (int e = 0; e < 4; ++e) {
            if (e == 0 && !n) continue;
            if (m.t8 && (e % 2)) continue;  // 8x8 transform: no luma edges inside the 8x8 blocks (odd edges carry no chroma either)
            if (mflat && e > 0) continue;
            if (mflat && e == 0) {
              const int nx = vertical ? mx - 1 : mx, ny = vertical ? my : my - 1;
              if (flat[(size_t)ny * W + nx] && same_motion(mx, my, nx, ny)) continue;
            }
            int bss[4];
            bool any = false;
            for (int k = 0; k < 4; ++k) {
              int qy4, qx4, py4, px4;
              if (vertical) qy4 = my * 4 + k, qx4 = mx * 4 + e, py4 = qy4, px4 = qx4 - 1;
              else qy4 = my * 4 + e, qx4 = mx * 4 + k, py4 = qy4 - 1, px4 = qx4;
              bss[k] = bs(py4, px4, qy4, qx4, e == 0);
              any = any || bss[k];
            }
            if (!any) continue;
            const int qp_p = e == 0 ? n->qp : m.qp;
            for (int plane = 0; plane < 3; ++plane) {
              if (plane && (e % 2)) continue;
              const bool luma = plane == 0;
              const int size = luma ? 16 : 8, stride = luma ? W * 16 : W * 8, PH = luma ? Hh * 16 : Hh * 8;
              uint8_t* P = luma ? cur->y : C(plane - 1);
              const int qpav = luma ? (qp_p + m.qp + 1) >> 1 : (qpc(qp_p, plane - 1) + qpc(m.qp, plane - 1) + 1) >> 1;
              const int idx_a = clip3(0, 51, qpav + off_a), idx_b = clip3(0, 51, qpav + off_b);
              const int alpha = ALPHA[idx_a], beta = BETA[idx_b];
              if (alpha == 0) continue;
              const int pos = luma ? e * 4 : e * 2;
              for (int k = 0; k < size; ++k) {
                const int bS = luma ? bss[k >> 2] : bss[k >> 1];
                if (bS == 0) continue;
                int px[8];
                int y, x;
                if (vertical) {
                  y = my * size + k, x = mx * size + pos;
                  for (int d = -4; d < 4; ++d) px[d + 4] = (x + d >= 0 && x + d < stride) ? P[(size_t)y * stride + x + d] : 0;
                } else {
                  y = my * size + pos, x = mx * size + k;
                  for (int d = -4; d < 4; ++d) px[d + 4] = (y + d >= 0 && y + d < PH) ? P[(size_t)(y + d) * stride + x] : 0;
                }
                if (!filter_line(px, bS, alpha, beta, idx_a, luma)) continue;
                for (int d = -3; d < 3; ++d) {
                  if (vertical) P[(size_t)y * stride + x + d] = (uint8_t)px[d + 4];
                  else P[(size_t)(y + d) * stride + x] = (uint8_t)px[d + 4];
                }
              }
            }
          }
        }
      }
  }

  // ---- slice_data (7.3.4)
  void run(const uint8_t* rbsp, int64_t n_bytes) {
    W = s->mb_w, Hh = s->mb_h, W4 = W * 4, H4 = Hh * 4;
    stype = s->slice_type, qp = s->qp;
    const int n_mb = W * Hh;
    mbs.assign(n_mb, MB());
    for (auto& m : mbs) m.typ = T_NONE;
    mvd.assign((size_t)2 * H4 * W4 * 2, 0);
    direct.assign((size_t)H4 * W4, 0), done.assign((size_t)H4 * W4, 0), nz.assign((size_t)H4 * W4, 0);
    tc.assign((size_t)H4 * W4, 0), tcc[0].assign((size_t)Hh * 2 * W * 2, 0), tcc[1].assign((size_t)Hh * 2 * W * 2, 0);
    // the current picture's motion data starts empty
    memset(cur->mv, 0, sizeof(int16_t) * 2 * H4 * W4 * 2);
    memset(cur->ref, 0xFF, (size_t)2 * H4 * W4);
    for (size_t i = 0; i < (size_t)2 * H4 * W4; ++i) cur->refid[i] = -1;
    memset(cur->intra4, 0, (size_t)H4 * W4);
    // (the planes need no clearing: every sample is written by a prediction before anything reads it)
    if (stype == 1 && s->weighted_mode == 2) {
      implicit.assign((size_t)s->nref[0] * s->nref[1] * 2, 32);
      for (int i = 0; i < s->nref[0]; ++i)
        for (int j = 0; j < s->nref[1]; ++j) {
          if (!l0[i].y || !l1[j].y) continue;
          const int tb = clip3(-128, 127, cur->poc - l0[i].poc), td = clip3(-128, 127, l1[j].poc - l0[i].poc);
          int w0 = 32, w1 = 32;
          if (td != 0) {
            const int tx = (16384 + std::abs(td / 2)) / td;
            const int dsf = clip3(-1024, 1023, (tb * tx + 32) >> 6);
            if ((dsf >> 2) >= -64 && (dsf >> 2) <= 128) w0 = 64 - (dsf >> 2), w1 = dsf >> 2;
          }
          implicit[((size_t)i * s->nref[1] + j) * 2] = w0, implicit[((size_t)i * s->nref[1] + j) * 2 + 1] = w1;
        }
    }
    bits.d = rbsp, bits.n_bits = n_bytes * 8, bits.p = s->data_bit_offset;
    if (cabac()) {
      cab.init(&bits, qp, stype == 2 ? CTX_I : CTX_PB0, stype == 2 ? CTX8_I : CTX8_PB0);
      for (int addr = 0; addr < n_mb; ++addr) {
        macroblock_cabac(addr, addr % W, addr / W);
        const int end = cab.terminate();
        H264_CHECK(end == (addr == n_mb - 1 ? 1 : 0), "end_of_slice_flag is not at the last macroblock: the CABAC decode lost synchronisation");
      }
      stats[4] = (int)(n_bytes * 8 - bits.p);
      H264_CHECK(stats[4] >= -16 && stats[4] <= 16, "slice data not used up at end_of_slice_flag");
    } else {
      H264_CHECK(stype != 1, "B slices with CAVLC entropy coding are not implemented");
      int64_t n = n_bytes;
      while (n && rbsp[n - 1] == 0) --n;
      H264_CHECK(n > 0, "empty slice payload");
      const int last = rbsp[n - 1];
      int tz = 0;
      while (!((last >> tz) & 1)) ++tz;
      stop_bit = (n - 1) * 8 + 7 - tz;
      int addr = 0;
      bool more = true;
      while (more) {
        if (stype != 2) {
          const int run = bits.ue();
          H264_CHECK(addr + run <= n_mb, "mb_skip_run beyond the picture");
          for (int i = 0; i < run; ++i, ++addr) skip_mb(addr, addr % W, addr / W);
          if (run) more = bits.p < stop_bit;
        }
        if (more) {
          H264_CHECK(addr < n_mb, "macroblock data beyond the picture");
          macroblock_cavlc(addr, addr % W, addr / W);
          ++addr;
          more = bits.p < stop_bit;
        }
      }
      H264_CHECK(addr == n_mb, "slice data ended before the last macroblock");
      stats[4] = (int)(stop_bit - bits.p);
      H264_CHECK(stats[4] == 0, "slice data not used up");
    }
    if (s->disable_deblock != 1) deblock();
  }
};

}  // namespace

extern "C" int sa_h264_decode_slice(const sa_h264_slice* s, const uint8_t* rbsp, int64_t n_bytes, const sa_h264_pic* list0,
                                    const sa_h264_pic* list1, sa_h264_pic* cur, int32_t* stats) {
  SA_REQUIRE(s && rbsp && cur && n_bytes > 0, "sa_h264_decode_slice: null argument");
  SA_REQUIRE(s->mb_w > 0 && s->mb_h > 0 && s->slice_type >= 0 && s->slice_type <= 2, "sa_h264_decode_slice: bad picture size or slice type");
  SA_REQUIRE(s->nref[0] >= 0 && s->nref[0] <= 32 && s->nref[1] >= 0 && s->nref[1] <= 32, "sa_h264_decode_slice: bad reference counts");
  SA_REQUIRE(s->slice_type == 2 || (list0 && s->nref[0] > 0), "sa_h264_decode_slice: a P / B slice needs RefPicList0");
  SA_REQUIRE(s->slice_type != 1 || (list1 && s->nref[1] > 0), "sa_h264_decode_slice: a B slice needs RefPicList1");
  SA_REQUIRE(cur->y && cur->cb && cur->cr && cur->mv && cur->ref && cur->refid && cur->intra4, "sa_h264_decode_slice: the current picture's buffers");
  SA_REQUIRE(s->data_bit_offset >= 0 && s->data_bit_offset < n_bytes * 8, "sa_h264_decode_slice: data_bit_offset");
  Slice sl;
  sl.s = s, sl.l0 = list0, sl.l1 = list1, sl.cur = cur;
  try {
    sl.run(rbsp, n_bytes);
  } catch (const Desync& e) {
    return sa::fail(SA_ERR_INVALID_ARG, "sa_h264_decode_slice: %s", e.what);
  }
  if (stats)
    for (int i = 0; i < 8; ++i) stats[i] = sl.stats[i];
  return SA_OK;
}

// Limited-range BT.601 YCbCr 4:2:0 -> BGR as libswscale's x86 SIMD path produces it (what cv2.VideoCapture hands to the reference's
// MediaVideo; io/_h264_intra.py `swscale_bgr` / `swscale_blue` are the same arithmetic in NumPy and this function's checker):
// 13-bit coefficients, truncating products of the samples shifted left by 3, chroma replicated 2 x 2, saturating pack.
extern "C" int sa_yuv420_to_bgr(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int width, int height, int y_stride, int c_stride,
                                uint8_t* out, int channels) {
  SA_REQUIRE(y && cb && cr && out && width > 0 && height > 0 && y_stride >= width && c_stride >= (width + 1) / 2 && (channels == 1 || channels == 3),
             "sa_yuv420_to_bgr: bad arguments");
  for (int j = 0; j < height; ++j) {
    const uint8_t *yr = y + (size_t)j * y_stride, *ur = cb + (size_t)(j >> 1) * c_stride, *vr = cr + (size_t)(j >> 1) * c_stride;
    uint8_t* o = out + (size_t)j * width * channels;
    for (int i = 0; i < width; ++i) {
      const int yy = ((((int)yr[i] - 16) * 8) * 9539) >> 16;
      const int u = ((int)ur[i >> 1] - 128) * 8, v = ((int)vr[i >> 1] - 128) * 8;
      const int b = yy + ((u * 16531) >> 16);
      if (channels == 1) {
        o[i] = (uint8_t)clip1(b);
      } else {
        const int g = yy - ((u * 3203) >> 16) - ((v * 6660) >> 16), r = yy + ((v * 13075) >> 16);
        o[3 * i] = (uint8_t)clip1(b), o[3 * i + 1] = (uint8_t)clip1(g), o[3 * i + 2] = (uint8_t)clip1(r);
      }
    }
  }
  return SA_OK;
}
