/* how many hardware threads does this process really get?  N busy threads for 0.5 s each round; prints work per thread relative to N = 1 */
#include <pthread.h>
#include <stdio.h>
#include <stdint.h>
#include <time.h>
static volatile int stop;
static void* spin(void* p) { uint64_t n = 0; while (!stop) { n++; __asm__ volatile("" ::: "memory"); } *(uint64_t*)p = n; return 0; }
int main(void)
{
    int ns[] = { 1, 8, 16, 32, 64, 128, 192, 256 };
    double base = 0;
    for (int k = 0; k < 8; k++)
    {
        int n = ns[k];
        pthread_t t[256]; uint64_t c[256];
        stop = 0;
        for (int i = 0; i < n; i++) pthread_create(&t[i], 0, spin, &c[i]);
        struct timespec ts = { 0, 500000000 }; nanosleep(&ts, 0);
        stop = 1;
        uint64_t sum = 0;
        for (int i = 0; i < n; i++) { pthread_join(t[i], 0); sum += c[i]; }
        if (!k) base = (double)sum;
        printf("threads %3d: total work %.1f x one thread (%.2f per thread)\n", n, sum / base, sum / base / n);
    }
    return 0;
}
