// XOR with a 2-4-1 sigmoid network, bce_loss and SGD(0.1) for 50 000 iterations -- counterpart of the
// reference's src/main.rs (the crate's demo binary), on the host library.
#include "common.h"

using namespace taper;

int main() {
    try {
        printf("XOR Neural Network Training\n\n");
        const std::vector<float> x_data = {0.f, 0.f, 0.f, 1.f, 1.f, 0.f, 1.f, 1.f}, y_data = {0.f, 1.f, 1.f, 0.f};   // main.rs:17-18
        auto model = std::make_shared<Sequential>(std::vector<std::shared_ptr<Module>>{
            std::make_shared<Linear>(2, 4, true, 1), std::make_shared<Sigmoid>(),                                    // main.rs:20-25
            std::make_shared<Linear>(4, 1, true, 2), std::make_shared<Sigmoid>()});
        SGD opt(model->parameters(), 0.10f);                                                                         // main.rs:28
        // the inputs never change: upload them once instead of per iteration (main.rs:35-36)
        const Tensor x(x_data, {4, 2}), y(y_data, {4, 1});
        const size_t epochs = 50000;
        for (size_t epoch = 0; epoch < epochs; ++epoch) {
            Tape::reset();
            Tensor loss = bce_loss(model->forward(x), y);
            loss.backward();
            opt.step();
            opt.zero_grad();
            if (epoch % 1000 == 0) printf("iteration %4zu: Loss = %.4f\n", epoch, loss.data()[0]);
        }
        Tape::reset();
        const std::vector<float> p = model->forward(x).data();
        printf("\n[0,0]->%.3f\n[0,1]->%.3f\n[1,0]->%.3f\n[1,1]->%.3f\n", p[0], p[1], p[2], p[3]);
        const bool ok = p[0] < 0.5f && p[1] > 0.5f && p[2] > 0.5f && p[3] < 0.5f;                                    // main.rs:66-67
        printf("%s\n", ok ? "learned XOR" : "not yet");
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
