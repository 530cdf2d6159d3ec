/* Test-infrastructure stub: main.cpp:16 includes this header but uses nothing from it. */
