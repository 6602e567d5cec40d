// gtest_lite: <gmock/gmock.h> — the reference's tests include it for ::testing::DoubleNear with ASSERT_THAT, which gtest.h here provides.
#ifndef UAMMD_TESTS_GMOCK_LITE_H
#define UAMMD_TESTS_GMOCK_LITE_H
#include "../gtest/gtest.h"
#endif
